"""CPU suite: the bench line kept under profiles/ carries every field of the bench.py contract (metric, value ...,
roofline, cpu_baseline), and bench.py's option surface is the one the driver uses."""
import json
import os
import re

from conftest import ROOT


def latest_line():
    names = sorted((f for f in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+_bench_v\d+\.json", f)),
                   key=lambda f: [int(x) for x in re.findall(r"\d+", f)])
    with open(os.path.join(ROOT, "profiles", names[-1])) as f:
        return names[-1], json.load(f)


def test_bench_line_has_the_contract_fields():
    name, d = latest_line()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, (name, k)
    assert d["unit"] == "modexp/s" and d["higher_is_better"] is True and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9 and 0 < r["frac"] < 1
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c.get("bit_exact_vs_gpu") is True
    # the value is elements / time of the timed steps
    assert abs(d["value"] - d["config"]["elements_per_gpu"] * d["n_gpus"] / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6


def test_bench_options():
    src = open(os.path.join(ROOT, "bench.py")).read()
    for opt in ("--gpus", "--steps", "--warmup"):
        assert f'"{opt}"' in src
    # the product legs never import the oracle: only the two cpu_baseline blocks do
    body = src.split("def main() -> None:")[1]
    head, tail = body.split("if rank == 0 and not args.no_cpu:")
    assert "oracle" not in head.replace('"oracle", "libvmnoracle.so"', "")


def _run_bench(args, env_extra, timeout=300):
    import subprocess
    import sys
    env = dict(os.environ, OMP_NUM_THREADS="1", **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                          timeout=timeout, env=env)


def test_bench_starts_its_own_ranks_when_typed_without_a_launcher(entry):
    """`python3 bench.py --gpus N` (the way the driver types the N = 1 command) must not exit with "use a launcher": the
    parent spawns `torch.distributed.run` as a child before touching the GPU, relays rank 0's line and its exit code."""
    pr = _run_bench(["--gpus", "3", "--steps", "2", "--warmup", "0", "--scaling", "strong"], {"VMN_BENCH_LAUNCH_ONLY": "1"})
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    lines = [ln for ln in pr.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, pr.stdout.decode()
    d = json.loads(lines[0])
    assert d == {"launch_only": True, "n_gpus": 3, "scaling": "strong", "steps": 2}
    # without a GPU the ranks refuse to run (no CPU fallback) and the parent hands their failure on
    pr = _run_bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], {})
    err = pr.stderr.decode()
    assert pr.returncode != 0 and "no GPU visible" in err and "must be launched" not in err
    assert "rehearsal" in err                          # fewer GPUs than ranks: the parent says what kind of run this is


def test_launched_rank_count_must_match_gpus():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "self_launch(args)" in src and "WORLD_SIZE" in src
    assert '"--scaling"' in src and "strong" in src
