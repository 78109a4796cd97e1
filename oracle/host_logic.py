"""Restatement of the reference's per-epoch controllers -- TEST INFRASTRUCTURE ONLY.

* ``PlateauLR``     : utils.LRScheduler (utils.py:285-323) = torch ReduceLROnPlateau(mode="min",
                      factor, patience, min_lr) with torch defaults threshold=1e-4 (rel), cooldown=0,
                      eps=1e-8.  Pinned by the reference's tests/test_utils.py:83-108 sequence.
* ``EarlyStop``     : utils.EarlyStopping (utils.py:248-282) including the equality quirk
                      (best - loss == min_delta changes nothing).
* ``split_indices`` : sklearn train_test_split(data, test_size, random_state=1) as used at
                      helper.py:315-317 (ShuffleSplit: permutation of RandomState(1); test = first
                      ceil(test_size*n) of it, train = the next floor((1-test_size)*n)).
"""
import math

import numpy as np


class PlateauLR:
    def __init__(self, lr, patience, min_lr=1e-6, factor=0.5, threshold=1e-4, eps=1e-8):
        self.lr = lr
        self.patience = patience
        self.min_lr = min_lr
        self.factor = factor
        self.threshold = threshold
        self.eps = eps
        self.best = math.inf
        self.num_bad = 0

    def step(self, metric):
        metric = float(metric)
        if metric < self.best * (1.0 - self.threshold):
            self.best = metric
            self.num_bad = 0
        else:
            self.num_bad += 1
        if self.num_bad > self.patience:
            new_lr = max(self.lr * self.factor, self.min_lr)
            if self.lr - new_lr > self.eps:
                self.lr = new_lr
            self.num_bad = 0
        return self.lr


class EarlyStop:
    def __init__(self, patience, min_delta):
        self.patience = patience
        self.min_delta = min_delta
        self.counter = 0
        self.best = None
        self.stop = False

    def step(self, loss):
        if self.best is None:
            self.best = loss
        elif self.best - loss > self.min_delta:
            self.best = loss
            self.counter = 0
        elif self.best - loss < self.min_delta:
            self.counter += 1
            if self.counter >= self.patience:
                self.stop = True
        return self.stop


def split_indices(n, test_size, random_state=1):
    n_test = int(math.ceil(test_size * n))
    n_train = int(math.floor((1.0 - test_size) * n))
    perm = np.random.RandomState(random_state).permutation(n)
    return perm[n_test:n_test + n_train], perm[:n_test]
