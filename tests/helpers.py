import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def gold_json():
    return json.load(open(os.path.join(GOLD, "golden.json")))


def load_gray(key):
    return np.load(os.path.join(GOLD, key + ".npz"))["gray"]


def n_scales(p):
    return (p.adaptiveThreshWinSizeMax - p.adaptiveThreshWinSizeMin) // p.adaptiveThreshWinSizeStep + 1
