"""Weight-init helpers with the reference names
(/root/reference/python/jdet/models/utils/weight_init.py)."""
import numpy as np
import torch.nn as nn


def normal_init(module, mean=0, std=1, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.normal_(module.weight, mean, std)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def constant_init(module, val, bias=0):
    if getattr(module, 'weight', None) is not None:
        nn.init.constant_(module.weight, val)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def xavier_init(module, gain=1, bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if getattr(module, 'weight', None) is not None:
        (nn.init.xavier_uniform_ if distribution == 'uniform' else nn.init.xavier_normal_)(module.weight, gain=gain)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def kaiming_init(module, a=0, mode='fan_out', nonlinearity='relu', bias=0, distribution='normal'):
    assert distribution in ['uniform', 'normal']
    if getattr(module, 'weight', None) is not None:
        fn = nn.init.kaiming_uniform_ if distribution == 'uniform' else nn.init.kaiming_normal_
        fn(module.weight, a=a, mode=mode, nonlinearity=nonlinearity)
    if getattr(module, 'bias', None) is not None:
        nn.init.constant_(module.bias, bias)


def bias_init_with_prob(prior_prob):
    return float(-np.log((1 - prior_prob) / prior_prob))
