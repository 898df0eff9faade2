from gen_adversarial_amd.experiments.alpha_learning.common_utils import (AlphaEvaluator, get_best_combination,  # noqa: F401
                                                                          get_cosine_alphas, get_linear_alphas, random_search)
