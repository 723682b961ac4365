#!/usr/bin/env python3
"""Paired comparison of two result files of tests/accuracy_experiment.py (same seeds): per-seed differences, paired t-test,
Welch's t-test and a sign test -- two means with overlapping spreads say nothing about a 2 400-step Adam trajectory.
usage: python tools/accuracy_stats.py A.json B.json [key=l1_visible_m]      (reports B - A)"""
import json
import sys

import numpy as np
from scipy import stats


def load(path, key):
    d = json.load(open(path))
    return {r["seed"]: r[key] for r in d["runs"]}, d["runs"][0].get("backend", "?")


def compare(pa, pb, key="l1_visible_m"):
    a, na = load(pa, key)
    b, nb = load(pb, key)
    seeds = sorted(set(a) & set(b))
    x, y = np.array([a[s] for s in seeds]), np.array([b[s] for s in seeds])
    d = y - x
    out = dict(key=key, A=pa, B=pb, backend_A=na, backend_B=nb, n=len(seeds), seeds=seeds,
               mean_A=float(x.mean()), sd_A=float(x.std(ddof=1)), mean_B=float(y.mean()), sd_B=float(y.std(ddof=1)),
               paired_diff_mean=float(d.mean()), paired_diff_sd=float(d.std(ddof=1)),
               paired_t_p=float(stats.ttest_rel(y, x).pvalue), welch_p=float(stats.ttest_ind(y, x, equal_var=False).pvalue),
               sign_test_p=float(stats.binomtest(int((d > 0).sum()), len(d), 0.5).pvalue), n_B_greater=int((d > 0).sum()),
               per_seed_diff=[round(float(v), 5) for v in d])
    return out


if __name__ == "__main__":
    key = sys.argv[3] if len(sys.argv) > 3 else "l1_visible_m"
    r = compare(sys.argv[1], sys.argv[2], key)
    print(json.dumps(r, indent=1))
