"""Competitor purification defenders of the reference (src/defenses/competitors): ND-VAE and A-VAE behind the same engine."""
