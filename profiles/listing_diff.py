"""Are the gfx950 instruction streams of the kernels the same as at a git revision?  (labels and symbol names aside)
Usage: python3 profiles/listing_diff.py [rev]   -- builds `make asm` of that revision in a temporary directory.
Used when a change is meant to leave existing kernels untouched (e.g. a new template parameter)."""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rev = sys.argv[1] if len(sys.argv) > 1 else "HEAD"

def bodies(path):
    out = {}
    for m in re.finditer(r"^(_ZN\w+):[^\n]*\n(.*?)\n\.Lfunc_end", open(path).read(), re.S | re.M):
        out[m.group(1)] = [re.sub(r"\.L\w+|_ZN\w+", "X", l.strip()) for l in m.group(2).splitlines()
                           if l.strip() and not l.strip().startswith((";", "."))]
    return out

with tempfile.TemporaryDirectory() as tmp:
    subprocess.run(f"git -C {ROOT} archive {rev} dbde-video-cpp_amd/csrc include | tar -x -C {tmp}", shell=True, check=True)
    subprocess.run(["make", "-s", "-C", f"{tmp}/dbde-video-cpp_amd/csrc", "asm"], check=True, capture_output=True)
    subprocess.run(["make", "-s", "-C", f"{ROOT}/dbde-video-cpp_amd/csrc", "asm"], check=True, capture_output=True)
    bad = 0
    for f in ("dbde_kernels.s", "dbde16_kernels.s"):
        old, new = bodies(f"{tmp}/dbde-video-cpp_amd/csrc/{f}"), bodies(f"{ROOT}/dbde-video-cpp_amd/csrc/{f}")
        for k, v in old.items():
            cands = [k, k.replace("EEEvNS_9EncParamsE", "ELi1EEEvNS_9EncParamsE"), k.replace("EEEvNS_9DecParamsE", "ELi256EEEvNS_9DecParamsE")]
            k2 = next((c for c in cands if c in new), None)
            if k2 is None:
                print("gone ", k); bad += 1
            elif new[k2] != v:
                print("DIFF ", k2, len(v), "->", len(new[k2])); bad += 1
        for k in new:
            if k not in old and k.replace("ELi1EEEvNS_9EncParamsE", "EEEvNS_9EncParamsE") not in old and \
                    k.replace("ELi256EEEvNS_9DecParamsE", "EEEvNS_9DecParamsE") not in old:
                print("new  ", k, len(new[k]))
    print("identical" if not bad else f"{bad} kernels differ")
