import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.fail("test marked gpu but no GPU is visible")
    # the CPU oracle runs beside the GPU: torch's intra-op pool oversubscribes badly on the 256-thread GPU hosts
    torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
    return torch.device("cuda:0")


# Parity notes: tests that grant near-tie exemptions record them through tests.util.parity_note; the terminal
# summary prints them right above pytest's last line, so that a `tail` of a quiet (-q) run still shows how many exemptions it used.
def pytest_terminal_summary(terminalreporter, exitstatus, config):
    from tests.util import PARITY_NOTES
    if PARITY_NOTES:
        total = sum(int(n.split("exempt=")[1].split()[0]) for n in PARITY_NOTES if "exempt=" in n)
        terminalreporter.write_line(f"parity notes ({len(PARITY_NOTES)}), near-tie exemptions used in this run: exempt={total}")
        for n in PARITY_NOTES:
            terminalreporter.write_line("  " + n)
