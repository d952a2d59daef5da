"""The reference's own module and the oracle port timed back to back on the same host cores (authoring container: /root/reference is mounted
here and absent on the GPU box, where bench.py's cpu_baseline therefore says kind "port").  Ties the port's clips/s to the reference's once:

    python tools/cpu_reference_baseline.py [--iters 3]   ->  profiles/r4_cpu_baseline_reference_vs_port.json

Workload of both legs: BASELINE configs[2] geometry, 1 clip 8 x 224^2, 52 visible tokens per frame (L = 417), fp32, forward + distillation
loss + backward (bench.py `_cpu_baseline_reference` and `cpu_baseline`, unchanged), alternating reference / port / reference / port."""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3)
    a = ap.parse_args()
    from internvideo_amd.hostinfo import usable_cores
    cores = usable_cores()
    torch.set_num_threads(cores)
    spec = bench.MODELS["1B"]
    legs = []
    for rnd in range(2):
        legs.append(bench._cpu_baseline_reference(spec, a.iters, cores))
        os.environ["IV_REFERENCE_ROOT"] = "/nonexistent"          # force the port leg of bench.cpu_baseline
        try:
            port = bench.cpu_baseline(spec, a.iters)
        finally:
            del os.environ["IV_REFERENCE_ROOT"]
        legs.append({k: port[k] for k in ("value", "unit", "cores", "kind", "sample")})
        print(json.dumps(legs[-2])); print(json.dumps(legs[-1]), flush=True)
    ref = [x["value"] for x in legs if x["kind"] == "reference"]; prt = [x["value"] for x in legs if x["kind"] == "port"]
    out = {"host": os.uname().nodename, "cores": cores, "torch": torch.__version__, "legs": legs,
           "reference_clips_per_s_mean": round(sum(ref) / len(ref), 4), "port_clips_per_s_mean": round(sum(prt) / len(prt), 4),
           "port_over_reference": round((sum(prt) / len(prt)) / (sum(ref) / len(ref)), 3),
           "note": "authoring container (no GPU); the GPU box has no reference tree, so BENCH_rNN lines carry kind 'port' on that box's cores"}
    path = os.path.join(ROOT, "profiles", "r4_cpu_baseline_reference_vs_port.json")
    json.dump(out, open(path, "w"), indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
