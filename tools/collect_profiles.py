#!/usr/bin/env python3
"""What `tools/measure.sh <tag>` left under gpurun_out/<tag>/ into profiles/ under the round's name:
    python tools/collect_profiles.py gpurun_out/r06c r06
Bench logs -> profiles/<round>_bench_<config>.json (the JSON line alone), the kernel statistics / PMC summaries the box made
(gpurun_out/<tag>/profiles/<tag>_*) renamed to <round>_*, phase clocks, stage comparison, stability (tracebacks of the digest mode's
warnings dropped), GPU suite, GN polish, the one-GPU multi-rank logic runs."""
import glob
import os
import shutil
import sys

src, rnd = sys.argv[1].rstrip("/"), sys.argv[2]
tag = os.path.basename(src)
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def first_json(path):
    if not os.path.exists(path):
        return None
    for line in open(path):
        if line.startswith("{"):
            return line
    return None


bench = {"bench_default.log": "bench_default.json", "bench_rs468.log": "bench_rs468.json", "bench_rs90.log": "bench_rs90.json", "bench_aerial.log": "bench_aerial.json",
         "bench_synth16k.log": "bench_synth16k.json", "bench_synth64k_16k.log": "bench_synth64k_16k.json",
         "bench_prof_nc3500.log": "bench_under_rocprof_nc3500.json", "bench_prof_rs468.log": "bench_under_rocprof_rs468.json",
         "bench_prof_synth16k.log": "bench_under_rocprof_synth16k.json", "bench_onegpu2_nc3500.log": "bench_onegpu2_logic_nc3500.json",
         "bench_onegpu8_nc3500.log": "bench_onegpu8_logic_nc3500.json", "bench_onegpu8_synth16k.log": "bench_onegpu8_logic_synth16k.json"}
for a, b in bench.items():
    line = first_json(os.path.join(src, a))
    if line is None:
        print("missing:", a)
        continue
    open(os.path.join(P, f"{rnd}_{b}"), "w").write(line)
plain = {"stage_compare_nc3500.txt": "stage_compare_nc3500.txt", "k9_phase.txt": "k9_phase_times.txt", "tr_phase.txt": "tr_phase_times.txt",
         "sn_phase_nc3500.txt": "sn_panel_phase_times.txt", "gpu_tests.log": "gpu_tests.txt", "gn_polish_nc3500.json": "gn_polish_nc3500.json",
         "gn_polish_rs468.json": "gn_polish_rs468.json", "gn_polish_rs90.json": "gn_polish_rs90.json"}
for a, b in plain.items():
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(P, f"{rnd}_{b}"))
    else:
        print("missing:", a)
if os.path.exists(os.path.join(src, "stab.txt")):
    keep = [line for line in open(os.path.join(src, "stab.txt")) if not line.startswith("Traceback") and not line.startswith("  File")]
    open(os.path.join(P, f"{rnd}_stability.txt"), "w").write("".join(keep))
for f in glob.glob(os.path.join(src, "profiles", f"{tag}_*")):
    b = os.path.basename(f).replace(f"{tag}_", f"{rnd}_", 1)
    if b.endswith(".json"):
        open(os.path.join(P, b), "w").write(open(f).read().replace(tag, rnd))
    else:
        shutil.copy(f, os.path.join(P, b))
print("done:", src, "->", P)
