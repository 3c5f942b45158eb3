"""Shared helpers for the parity tests: load a golden case and rebuild its seeded inputs."""
import ast
import os

import numpy as np
import torch

from openglue_amd import synthetic as syn

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MATCH_THRESHOLD = 0.2


def load_case(name):
    """-> (golden arrays, config, state_dict, data) for tests/golden/<name>.npz."""
    z = dict(np.load(os.path.join(GOLDEN, f"{name}.npz"), allow_pickle=False))
    kw = ast.literal_eval(str(z["config_kwargs"]))
    cfg = syn.make_config(**kw)
    m, n, batch, seed = int(z["m"]), int(z["n"]), int(z["batch"]), int(z["seed"])
    if any(k.startswith("sd_") for k in z):      # case stored with its inputs/weights in full
        sd = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("sd_")}
        data = {k[3:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("in_")}
        data["image0_size"] = list(syn.IMAGE_WH)
        data["image1_size"] = list(syn.IMAGE_WH)
    else:
        sd = syn.make_state_dict(cfg, seed=0)
        data = syn.make_batch(batch, m, n, cfg["descriptor_dim"],
                              cfg["positional_encoding"]["side_info_size"], seed=seed)
    return z, cfg, sd, data


def to_device(data, device):
    return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in data.items()}


PARITY_NOTES = []      # printed by conftest.pytest_terminal_summary


def parity_note(text: str):
    PARITY_NOTES.append(text)
    print(text)
