#!/usr/bin/env python
"""tools/restamp_profile_sha.py COMMIT — one-off of round 3: salva_amd.kernel_source_sha() used to hash every file under
salva_amd/csrc; it now leaves out the exchange transports and the extern "C" shims (no profiled kernel lives there), so that
adding a transport does not orphan the committed single-GPU profiles.  This script re-stamps profiles/r03_*/hbm_traffic.json
with the hash under the new definition — but only after checking, from git, that the hashed files at COMMIT (the commit the
profiles were taken on) are byte-identical to the working tree's; the old stamp is kept as `kernel_src_sha_all_files`."""
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from salva_amd import kernel_source_sha  # noqa: E402

SKIP = {"comm.h", "comm.hip", "comm_peer.hip", "capi.hip"}


def sha_at(commit: str) -> str:
    names = subprocess.run(["git", "ls-tree", "--name-only", f"{commit}:salva_amd/csrc"], cwd=ROOT, check=True, capture_output=True,
                           text=True).stdout.split()
    hsh = hashlib.sha256()
    for n in sorted(n for n in names if n.endswith((".hip", ".h"))):
        if n in SKIP:
            continue
        hsh.update(n.encode())
        hsh.update(subprocess.run(["git", "show", f"{commit}:salva_amd/csrc/{n}"], cwd=ROOT, check=True, capture_output=True).stdout)
    return hsh.hexdigest()[:16]


def main():
    commit = sys.argv[1]
    then, now = sha_at(commit), kernel_source_sha()
    print("kernel sources at", commit, ":", then, "| tree:", now)
    if then != now:
        raise SystemExit("the kernel sources changed since the profiles were taken: re-profile instead (tools/profile_r03.sh)")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r03_*", "hbm_traffic.json"))):
        j = json.load(open(f))
        if j.get("kernel_src_sha") == now:
            continue
        j["kernel_src_sha_all_files"] = j.get("kernel_src_sha")
        j["kernel_src_sha"] = now
        j["kernel_src_sha_note"] = (f"re-stamped by tools/restamp_profile_sha.py: the hash now leaves out comm.h / comm.hip / comm_peer.hip / "
                                    f"capi.hip; the hashed files are byte-identical to those of commit {commit}, on which this profile was taken")
        json.dump(j, open(f, "w"), indent=1)
        print("re-stamped", os.path.relpath(f, ROOT))


if __name__ == "__main__":
    main()
