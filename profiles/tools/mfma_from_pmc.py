"""MFMA-pipe utilisation per contraction kernel class from the PMC pass of collect_mfma.sh:
busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs) over SQ_BUSY_CU_CYCLES (both summed over CUs);
issue-stalled / parked = SQ_WAIT_INST_ANY / SQ_WAIT_ANY over SQ_WAVE_CYCLES."""
import collections
import csv
import glob
import json
import sys



def _digest():
    import os
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
    from denoising_diffusion_deep_fake_amd import _lib
    return _lib.source_digest()

def main(out_dir, json_path, head="unknown", date="", bench_args=""):
    path = glob.glob(f"{out_dir}/**/*counter_collection.csv", recursive=True)[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        cls = ("conv_igemm" if "conv_igemm" in name else "conv_winograd" if "conv_winograd" in name
               else "conv_wgrad_patch" if "conv_wgrad_patch" in name
               else "conv_wgrad" if "conv_wgrad_kernel" in name else "conv_patch" if "conv_patch_kernel" in name
               else None)
        if cls is None:
            continue
        agg[cls][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES":
            n[cls] += 1
    rows = []
    for cls, c in agg.items():
        busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 4.0 / max(c["SQ_BUSY_CU_CYCLES"], 1.0)
        rows.append({"kernel": cls, "launches": n[cls], "mfma_pipe_busy": round(busy, 3),
                     "wave_cycles_issue_stalled": round(c["SQ_WAIT_INST_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3),
                     "wave_cycles_parked": round(c["SQ_WAIT_ANY"] / max(c["SQ_WAVE_CYCLES"], 1), 3),
                     "mfma_mops_f32": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32"),
                     "mfma_mops_bf16": c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16")})
        print(rows[-1])
    json.dump({"source": "profiles/tools/collect_mfma.sh (6 training steps incl. warm-up, 256x256 bs16 f32 unless "
                         "bench_args says otherwise; kernels serialised by PMC collection)", "bench_args": bench_args,
               "git_head": head, "csrc_digest": _digest(), "date": date, "kernels": rows},
              open(json_path, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:6])
