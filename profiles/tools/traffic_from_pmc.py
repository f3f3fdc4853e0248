"""Per-launch HBM traffic of the forward conv_igemm launches from the two PMC passes of collect_traffic.sh.

Forward launches = the first 47 conv_igemm_kernel dispatches after each noise_blend_kernel dispatch (one training
step: blend -> forward -> loss -> backward -> Adam).  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950
FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section)."""
import csv
import glob
import json
import sys

N_FWD = 47


def per_launch(counter_dir, counter):
    path = glob.glob(f"{counter_dir}/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    vals, left = [], 0
    for r in rows:
        name = r["Kernel_Name"]
        if "noise_blend_kernel" in name:
            left = N_FWD
        elif "conv_igemm_kernel" in name and left > 0:
            vals.append(float(r["Counter_Value"]))
            left -= 1
    return vals


def main(out_dir, json_path):
    f = per_launch(f"{out_dir}/FETCH_SIZE", "FETCH_SIZE")
    w = per_launch(f"{out_dir}/WRITE_SIZE", "WRITE_SIZE")
    assert f and len(f) == len(w) and len(f) % N_FWD == 0, (len(f), len(w))
    fm, wm = sum(f) / len(f), sum(w) / len(w)
    res = {"kernel": "conv_igemm_kernel (forward launches)", "launches_sampled": len(f),
           "fetch_size_kb_mean": round(fm, 1), "write_size_kb_mean": round(wm, 1), "fetch_correction": 2.0,
           "hbm_bytes_per_launch": int(round((2.0 * fm + wm) * 1024)),
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, --kernel-trace) on `bench.py "
                     "--steps 4 --warmup 2 --no-cpu-baseline --no-kernel-events --no-alt`, 256x256 bs16 f32; bytes = "
                     "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section; "
                     "profiles/tools/collect_traffic.sh"}
    json.dump(res, open(json_path, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
