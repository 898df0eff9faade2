from gen_adversarial_amd.experiments.test_defense import *  # noqa: F401,F403
from gen_adversarial_amd.experiments.test_defense import main, parse_args, run_worker  # noqa: F401

if __name__ == '__main__':
    main()
