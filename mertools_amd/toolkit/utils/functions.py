"""The two config helpers main-release.py uses — mirror of MERBench/toolkit/utils/functions.py:144-159."""
import argparse
import random


def merge_args_config(args, config):
    """Copy config keys into args only where args lacks them or holds None."""
    args_dic = vars(args)
    for key in config:
        if key not in args_dic or args_dic[key] is None:
            args_dic[key] = config[key]
    return argparse.Namespace(**args_dic)


def func_random_select(config):
    """One random value per hyper-parameter list (python `random`, as the reference; seed it for repeatability)."""
    for key in config:
        values = config[key]
        config[key] = values[random.randint(0, len(values) - 1)]
    return config
