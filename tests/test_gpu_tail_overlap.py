"""The tail of a post-correction chunk must step BESIDE its last round whatever the host process did before the library's first call
(round-5 review, item 2; src/approx_mcmc.cpp:364-369 spreads the same loop over OpenMP threads).  Each state runs in a process of its own
(the arrangement of a process's streams is fixed by its history): `fresh`, and `foreign_plan` -- sixteen streams created and used by the
host (torch) and a full-output filter batch, both before the first post-correction call.  The 293-draw shard of C5 (one rank's share at
eight ranks: 256 workgroup-resident filters + 37 through the row family) is timed singly and over the logical shards [0, 0, 0, 0]."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _probe(state):
    env = {k: v for k, v in os.environ.items() if not k.startswith("BSSM_AMD_")}
    cp = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "tail_queue_probe_r06.py"), state, "--short"], cwd=ROOT, env=env,
                        capture_output=True, text=True, timeout=600)
    assert cp.returncode == 0, cp.stderr[-2000:]
    line = [ln for ln in cp.stdout.splitlines() if ln.startswith("JSON ")]
    assert len(line) == 1, cp.stdout[-2000:]
    return json.loads(line[0][5:])


@pytest.mark.parametrize("state", ["fresh", "foreign_plan"])
def test_the_c5_shard_overlaps_its_tail_in_every_process_state(state):
    r = _probe(state)
    # the stream that carries the tail was MEASURED to run beside the null stream's round (csrc/options.cpp: prewarm_streams)
    assert r["overlap_state"] == 1 and 1 <= r["candidates"] <= 8, r
    # 5.6-5.8 ms with the overlap; 7.1 ms with the tail behind its round (profiles/r06_tail_queue.txt); the review's bar is 6.2 ms
    assert r["single_device_phase_ms"] <= 6.2, r
    # the same job over four logical shards of the one GPU: the same bits (the draws are keyed by their global index)
    assert r["identical"], r
