"""Shim of the timm.utils names fastervit/validate.py imports (validate.py:29-30)."""
import argparse
import ast
import logging
import re


def accuracy(output, target, topk=(1,)):
    maxk = min(max(topk), output.size()[1])
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.reshape(1, -1).expand_as(pred.t()))
    return [correct[:min(k, maxk)].reshape(-1).float().sum(0) * 100. / target.size(0) for k in topk]


class AverageMeter:
    def __init__(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def natural_key(s):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s.lower())]


def setup_default_logging(default_level=logging.INFO, log_path=""):
    logging.basicConfig(level=default_level)


def set_jit_fuser(fuser):
    pass


def decay_batch_step(batch_size, num_intra_steps=2, no_odd=False):
    return max(batch_size // 2, 0)


def check_batch_size_retry(error_str):
    return False


class ParseKwargs(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        kw = {}
        for value in values:
            key, value = value.split("=")
            try:
                kw[key] = ast.literal_eval(value)
            except (ValueError, SyntaxError):
                kw[key] = str(value)
        setattr(namespace, self.dest, kw)
