"""profiles/rNN_pmc_counters.md -> profiles/rNN_traffic.json: HBM-side bytes per launch of the dominant kernel (FETCH_SIZE x 2, the
gfx950 correction of MI355X_MICROARCH.md confirmed by the calibration pass at the end of the same file, + WRITE_SIZE), stamped with the
sha256 of the kernel's source so that bench.py can tell when the record has gone stale.   usage: python tools/make_traffic_json.py r03 [kernel-name-prefix]"""
import hashlib, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
KERN = sys.argv[2] if len(sys.argv) > 2 else "match_mx6_screen_w4_kernel<256, 8"
md = open(os.path.join(ROOT, "profiles", f"{tag}_pmc_counters.md")).read()
def per_dispatch(kernel, counter):
    m = re.search(r"\| `[^`]*" + re.escape(kernel) + r"[^`]*` \| " + counter + r" \| [^|]+\| (\d+) \| ([^|]+)\|", md)
    return float(m.group(2))
fetch_kib, write_kib = per_dispatch(KERN, "FETCH_SIZE"), per_dispatch(KERN, "WRITE_SIZE")
sha = hashlib.sha256(open(os.path.join(ROOT, "oryon_amd", "csrc", "match16.hip" if "i8" in KERN else "screen_mx6.hip"), "rb").read()).hexdigest()
rec = {"kernel": KERN + ">", "workload": "cfg2: B=64, 224x224, C=256, NCHW", "fetch_size_kib_per_launch": fetch_kib,
       "fetch_correction": 2.0, "write_size_kib_per_launch": write_kib, "traffic_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
       "kernel_source_sha256": sha,
       "source": f"profiles/{tag}_pmc_counters.md (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE x 2: the calibration kernel in "
                 "the same file reads 1 GiB and the counter reports 0.5 GiB)"}
json.dump(rec, open(os.path.join(ROOT, "profiles", f"{tag}_traffic.json"), "w"), indent=1)
print(rec)
