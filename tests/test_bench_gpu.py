"""GPU suite: bench.py's contract line, N = 1 and the N > 1 code path (index broadcast, per-rank read shards, barrier +
max-over-ranks timing).  A 1-GPU box cannot run two RCCL ranks, so the N = 2 case uses the bench's own test hook
MOVI_BENCH_SHARE_GPU=1: both ranks on cuda:0, collectives over gloo, everything else as in production."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _line(out):
    lines = [l for l in out.decode().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tiny", "--steps", "3", "--warmup", "1"],
                       capture_output=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _line(r.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1                   # the contract's label (an HBM-bandwidth roofline) ...
    assert d["roofline"]["gathers_served_from"].startswith("infinity cache")                  # ... 1.6 MB of rows (3.2 MB as walked): where the gathers are served from follows the rows they walk
    assert d["roofline"]["side_table_bytes"] == 256 << 20                                      # ... the top-of-walk table is reported beside it
    assert d["roofline"]["lane_iterations_per_s"] > 0 and d["roofline"]["rows_read_per_s"] > 0
    assert d["roofline"]["kernel"].startswith("pml_kernel_flatp<6, unsigned int, 0, 0, 0, 1, ") and d["rccl_ranks"] == 0
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["parity_sample_ok"] is True


@pytest.mark.parametrize("query", ["pml", "count"])
def test_bench_two_ranks_share_one_gpu(query):
    env = dict(os.environ, MOVI_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533" if query == "pml" else "29534", os.path.join(ROOT, "bench.py"), "--gpus", "2",
           "--workload", "tiny", "--steps", "3", "--warmup", "1", "--query", query]
    r = subprocess.run(cmd, capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["bases_per_step_per_gpu"] == 20000 * 150
    assert "cpu_baseline" not in d                      # rank 0, N == 1 only


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started by hand (no torchrun): the parent spawns the ranks itself and relays rank 0's line."""
    env = dict(os.environ, MOVI_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "3",
                        "--warmup", "1"], capture_output=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["bases_per_step_per_gpu"] == 20000 * 150
    assert d["rccl_ranks"] == 0 and "index_broadcast_s" in d      # shared-GPU hook: gloo; under RCCL rccl_ranks == n_gpus


def test_bench_two_ranks_print_all_three_legs():
    """The N > 1 line carries, next to the headline, the long-read leg (plain and with --classify fused) and the big-table leg
    (rank 0 synthesises the table and draws every rank's reads; the rows reach the other rank by broadcast), each with
    per-rank seconds and a roofline object.  Shrunk: an 8 x 60 kbp pangenome, 2 M random rows, 256 long reads."""
    env = dict(os.environ, MOVI_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tinypg", "--steps", "3",
                        "--warmup", "1", "--big-rows", "2000000", "--long-reads", "256", "--no-sustained"],
                       capture_output=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and len(d["rank_seconds"]) == 2
    lr, bt = d["long_reads"], d["big_table"]
    assert lr["n_gpus"] == 2 and lr["value"] > 0 and len(lr["rank_seconds"]) == 2 and lr["errors"] == 0
    assert lr["roofline"]["frac"] > 0 and lr["classify_bins_only"]["roofline"]["frac"] > 0 and lr["classify_bins_agree"] is True
    assert lr["classify_vector_and_bins"]["fused_classify"] == 1 and lr["classify_bins_only"]["fused_classify"] == 2
    assert bt["n_gpus"] == 2 and bt["rows"] == 2000000 and bt["value"] > 0 and bt["parity_sample_ok"] is True
    assert bt["roofline"]["frac"] > 0 and bt["count"]["roofline"]["frac"] > 0 and bt["index_broadcast_s"] >= 0
    assert "cpu_baseline" not in d


def test_bench_default_line_legs_on_one_gpu():
    """The default line's host-side legs, shrunk (8 x 60 kbp pangenome, 2 M random rows, 256 long reads): `cli_path` -- the `movi query`
    binary on FASTA files, also with `--gpus 2` sharing the box's device --, `host_path`, and the same two for the big table
    (round 5: `big_table.host_path`, `big_table.cli_path` on the table's own index file)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "tinypg", "--steps", "3", "--warmup", "1",
                        "--big-rows", "2000000", "--long-reads", "256", "--no-sustained"], capture_output=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["parity_sample_ok"] is True and d["cpu_baseline"]["value"] > 0
    cp = d["cli_path"]
    assert "error" not in cp, cp
    for k in ("short_1Mx150_no_output", "short_1Mx150_bpf", "short_1Mx150_no_output_gpus2_one_device"):
        assert cp[k]["value"] > 0 and cp[k]["stage_s"]["parse"] > 0, (k, cp[k])
    bt = d["big_table"]
    assert bt["parity_sample_ok"] is True and bt["prepare_s"] >= 0 and bt["derived_bytes"] > 0
    assert bt["host_path"]["pageable"] > 0 and bt["host_path"]["pageable_ok"] is True and bt["host_path"]["page_locked_ok"] is True, bt["host_path"]
    assert bt["cli_path"]["no_output"]["value"] > 0 and bt["cli_path"]["bpf"]["value"] > 0, bt["cli_path"]
    assert bt["count"]["kernel"].startswith("zml_kernel_flat<6, unsigned int, 0, 0, ") and bt["count"]["parity_sample_ok"] is True


def test_bench_c4_two_ranks_share_one_gpu():
    """--workload c4 at N = 2: only rank 0 synthesises the table (and draws both shards of reads)."""
    env = dict(os.environ, MOVI_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "c4", "--rows", "20000000",
                        "--reads", "100000", "--steps", "3", "--warmup", "1"], capture_output=True, timeout=1200, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["rows"] == 20000000
    assert d["config"]["bases_per_step_per_gpu"] == 100000 * 150
