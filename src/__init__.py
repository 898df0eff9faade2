"""Import-path compatibility: the reference is imported as `src.defenses...` / `src.experiments...`; these modules
re-export the MI355X implementations so that caller code written against the reference keeps its import lines."""
