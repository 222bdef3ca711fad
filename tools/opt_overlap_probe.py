#!/usr/bin/env python3
"""Upper bound of what running the optimizer pass beside the next step's forward could buy (tools/ only, an experiment: the weights
race on purpose - the side stream is never joined - so the loss printed is NOT a result; only ms_per_step is read).
usage: opt_overlap_probe.py [bench.py arguments]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import importlib
import bench
training = importlib.import_module("graph-gpt_amd.training")
_orig = training.GgetEngine.step
_side = []


def step(self):
    if not _side:
        _side.append(torch.cuda.Stream())
    side = _side[0]
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        return _orig(self)


training.GgetEngine.step = step
bench.main()
