"""Copies the summaries tools/profile_r5.sh left in gpurun_out/prof_r5/ into profiles/ and merges their stamps into
profiles/r5_stamp.json (entries of workloads that were not profiled again stay as they are).  `python tools/merge_profiles.py`."""
import json
import os
import shutil

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof_r5")
DST = os.path.join(ROOT, "profiles")


def main():
    with open(os.path.join(SRC, "r5_stamp.json")) as f:
        new = json.load(f)
    path = os.path.join(DST, "r5_stamp.json")
    with open(path) as f:
        stamp = json.load(f)
    for name, entry in new.items():
        src = os.path.join(SRC, os.path.basename(name))
        if os.path.exists(src):
            shutil.copy(src, os.path.join(DST, os.path.basename(name)))
            stamp[name] = entry
    for extra in os.listdir(SRC):
        if extra.startswith("r5_") and extra != "r5_stamp.json" and os.path.isfile(os.path.join(SRC, extra)):
            shutil.copy(os.path.join(SRC, extra), os.path.join(DST, extra))
    with open(path, "w") as f:
        json.dump(stamp, f, indent=1, sort_keys=True)
    print(sorted(new))


if __name__ == "__main__":
    main()
