"""The end-of-task exchange over RCCL itself (backend "nccl" on ROCm), on device memory: `engine/evaluate.py: gather_answers` does one
`all_reduce(MAX)` of the record width and ONE `all_gather_into_tensor` of the fixed-width int32 records.  The CPU suite runs it with
gloo at world size 2; a gpurun box has one GPU, so this runs a ONE-rank RCCL process group - the same calls on device tensors, the
rank-0 rebuild of every record from the gathered buffer - and requires the files to be byte-identical to the run without a process group."""
import os
import re
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def _run(tmp_path, tag, extra):
    out = tmp_path / tag
    env = {**os.environ, "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29655", **extra}
    res = subprocess.run([sys.executable, str(ROOT / "tests" / "dist_worker.py"), str(out)], env=env, cwd=str(ROOT), capture_output=True,
                         text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    mask = lambda t: re.sub(r" at 0x[0-9a-f]+", "", re.sub(r'"(end_time|total_evaluation_time_seconds)": [^\n]*', r'"\1": 0', t))  # noqa: E731
    return {p.name: mask(p.read_text()) for p in sorted(out.rglob("*")) if p.is_file()}


@pytest.mark.parametrize("tokens", ["0", "1"])
def test_record_gather_over_rccl_matches_no_process_group(gpu, tmp_path, tokens):
    plain = _run(tmp_path, f"plain{tokens}", {"OWC_TEST_TOKENS": tokens})
    rccl = _run(tmp_path, f"rccl{tokens}", {"OWC_TEST_TOKENS": tokens, "OWC_TEST_BACKEND": "nccl"})
    assert len(plain) == 2 and plain == rccl
