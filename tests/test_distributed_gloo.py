"""CPU suite: the multi-GPU path's host logic on two (and three, ragged) gloo ranks.  The sharded proof of
a shuffle must reproduce the single-process oracle transcript exactly, accept it, and reject a reply
tampered on one rank only."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def run_world(world, backend, bits, n, width, tmp_path, timeout=600, flow="pos"):
    out = tmp_path / f"dist_{backend}_{flow}_{world}_{n}.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), backend, str(bits), str(n), str(width), str(out), flow]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=env)
    assert proc.returncode == 0, proc.stdout.decode()[-3000:]
    return json.load(open(out))


def run_cases(world, cases, tmp_path, timeout=900):
    """Several cases (backend, bits, n, width, flow) in ONE launch of `world` ranks (tests/dist_worker.py --cases): the
    ranks' start-up costs more than most cases.  Returns the cases' results in order."""
    outs, argv = [], []
    for k, (backend, bits, n, width, flow) in enumerate(cases):
        outs.append(tmp_path / f"dist_{k}_{backend}_{flow}_{world}_{n}.json")
        argv.append([backend, str(bits), str(n), str(width), str(outs[-1]), flow])
    spec = tmp_path / f"cases_{world}_{len(cases)}.json"
    spec.write_text(json.dumps(argv))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "dist_worker.py"), "--cases", str(spec)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=timeout, env=env)
    assert proc.returncode == 0, proc.stdout.decode()[-3000:]
    return [json.load(open(o)) for o in outs]


def test_shard_bounds_cover_everything(entry):
    import importlib.util
    import mirror
    par = mirror.load(entry, ("parallel",))["parallel"]
    for n in (0, 1, 7, 8, 9, 1000003):
        for world in (1, 2, 3, 8):
            spans = [par.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1


def test_host_row_blocks_of_every_kind_are_sliced_alike(entry):
    """The shared tape may hand out rows as bytes, as a (page-locked) host tensor or as a uint8 array: a shard takes
    the same rows from each."""
    import importlib.util
    import numpy as np
    import torch
    import mirror
    par = mirror.load(entry, ("parallel_mirror",))["parallel_mirror"]
    nb, n = 32, 50
    raw = bytes(range(256)) * (nb * n // 256 + 1)
    raw = raw[: nb * n]
    kinds = [raw, torch.frombuffer(bytearray(raw), dtype=torch.uint8), np.frombuffer(raw, dtype=np.uint8)]
    idx = [7, 3, 49, 0, 3]
    for block in kinds:
        assert par._take_rows(block, range(10, 20), nb) == raw[10 * nb:20 * nb]
        assert par._take_rows(block, idx, nb) == b"".join(raw[i * nb:(i + 1) * nb] for i in idx)
    ints = [int.from_bytes(raw[i * nb:(i + 1) * nb], "big") for i in range(n)]
    assert par._take_rows(ints, idx, nb) == [ints[i] for i in idx]


def test_comm_falls_back_to_gloo_when_the_device_path_raises(tmp_path):
    """bench.py hands parallel.Comm a gloo group as a safety net for the never-yet-executed RCCL branch: when the device
    all-gather raises, the communicator switches to it for good, records why, and the exchanges stay correct."""
    res = run_world(2, "fake", 512, 4, 1, tmp_path, flow="comm-fallback")
    assert res["pass"], res["why"]


@pytest.mark.parametrize("world,n,width", [(2, 21, 1), (3, 10, 2)])
def test_sharded_pos_matches_oracle_on_gloo(world, n, width, tmp_path):
    res = run_world(world, "fake", 512, n, width, tmp_path)
    assert res["pass"], res["why"]


def test_sharded_pos_over_p256_on_gloo(tmp_path):
    """The sharded driver is group-agnostic: the same exchanges carry curve points (BASELINE configs[4])."""
    res = run_world(2, "fake-ec", 256, 7, 1, tmp_path)
    assert res["pass"], res["why"]


def test_more_ranks_than_elements(tmp_path):
    """Ragged extreme: some ranks own an empty shard."""
    res = run_world(3, "fake", 512, 2, 1, tmp_path)
    assert res["pass"], res["why"]


@pytest.mark.parametrize("world,n,width,backend", [(2, 13, 1, "fake"), (3, 8, 2, "fake"), (3, 2, 1, "fake"), (2, 6, 3, "fake-ec")])
def test_sharded_ccpos_matches_oracle_on_gloo(world, n, width, backend, tmp_path):
    """BASELINE configs[4] is a CCPoS over P-256 at width 3 on 8 GPUs: the sharded commitment-consistent proof, plain and
    raised verifier, ragged and empty shards included, against the single-process oracle."""
    res = run_world(world, backend, 512 if backend == "fake" else 256, n, width, tmp_path, flow="ccpos")
    assert res["pass"], res["why"]


@pytest.mark.gpu
def test_sharded_cxx_drivers_two_ranks_one_gpu(tmp_path):
    """The sharded C++ drivers (vmn_pos / vmn_posc / vmn_ccpos with a communicator, include/vmnproofs.h) on the real
    kernels: two ranks share the one GPU of the test box, gloo carries the all-gather callback (on an 8-GPU node the
    backend is nccl = RCCL).  P-256 at width 3 is BASELINE configs[4]'s shape.  (CCPoS over a modular group at width 2: the
    three-rank test and the seeded 2048-bit case below.)  Four cases, one launch."""
    cases = [("hip-gloo", 2048, 150, 1, "pos"), ("hip-gloo", 512, 77, 1, "posc"), ("hip-gloo-ec", 256, 40, 3, "ccpos"),
             ("hip-gloo-ec", 256, 60, 1, "pos")]
    for case, res in zip(cases, run_cases(2, cases, tmp_path)):
        assert res["pass"], (case, res["why"])
        assert all(x is None or x <= 12 for x in res["exchanges"]), (case, res["exchanges"])      # a handful of exchanges per proof


@pytest.mark.gpu
@pytest.mark.parametrize("world,cases", [(2, [("hip-gloo", 2048, 131, 1, "pos-seeded"), ("hip-gloo-ec", 256, 50, 3, "ccpos-seeded")]),
                                         (3, [("hip-gloo", 512, 5, 2, "pos-seeded"), ("hip-gloo", 2048, 64, 1, "ccpos-seeded")])])
def test_sharded_cxx_drivers_generate_only_their_rows_of_the_prg_arrays(world, cases, tmp_path):
    """The form bench.py's multi-GPU legs run: r, s, b, beta, epsilon and the batching vector are 32-byte seeds, and a rank
    expands just its positions and the rows it reads through the permutation (struct Draw, csrc/vmnproofs.cpp;
    vmn_shuffle_reencrypt_shard_seeded, vmn_permutation_commitment_shard_seeded).  Transcript, u, w', r, s shards == the
    oracle run on the fully expanded arrays."""
    for case, res in zip(cases, run_cases(world, cases, tmp_path)):
        assert res["pass"], (case, res["why"])


@pytest.mark.gpu
def test_sharded_cxx_drivers_over_rccl_world_size_one(tmp_path):
    """The RCCL branch of parallel.Comm on real hardware: torch.distributed backend `nccl` (= RCCL), world size 1 on the test
    box's GPU (RCCL needs one GPU per rank; the box has one), communicator built as bench.py builds it (device + gloo safety
    net, so the constructor's probe exchange runs).  The sharded C++ drivers then run a PoS and a seeded CCPoS through it:
    every all-gather callback goes pinned host -> device -> all_gather_into_tensor -> pinned host.  The transcript must be
    the oracle's, the transport must be RCCL, and the fallback must not have been taken."""
    cases = [("hip", 2048, 96, 1, "pos"), ("hip", 2048, 64, 1, "ccpos-seeded")]
    for case, res in zip(cases, run_cases(1, cases, tmp_path)):
        assert res["pass"], (case, res["why"])
        assert res["comm"]["torch_backend"] == "nccl", res["comm"]
        assert res["comm"]["backend_used"] == "nccl" and res["comm"]["fell_back"] is None, res["comm"]
        assert all(x is None or x > 0 for x in res["exchanges"]), res["exchanges"]


@pytest.mark.gpu
def test_sharded_cxx_drivers_three_ranks_ragged_and_empty(tmp_path):
    cases = [("hip-gloo", 512, 2, 1, "pos"), ("hip-gloo", 512, 10, 2, "ccpos")]          # (the first: one rank owns nothing)
    for case, res in zip(cases, run_cases(3, cases, tmp_path)):
        assert res["pass"], (case, res["why"])


@pytest.mark.gpu
def test_sharded_pos_real_kernels_two_ranks_one_gpu(tmp_path):
    """The same sharded proof with the HIP library doing the arithmetic: two ranks share the one GPU of
    the test box (gloo carries the small exchanges; on an 8-GPU node the backend is nccl = RCCL)."""
    cases = [("hip-gloo-mirror", 2048, 150, 1, "pos"), ("hip-gloo-ec-mirror", 256, 60, 1, "pos")]     # the Python mirror on the real kernels; P-256
    for case, res in zip(cases, run_cases(2, cases, tmp_path)):
        assert res["pass"], (case, res["why"])
