"""profiles/rNN_pmc_traffic.json from the per-kernel PMC averages tools/validate_gpu.sh collects (gpurun_out/val_pmc_hbm_counters.txt = the output
of tools/pmc_summary.py over the FETCH_SIZE / WRITE_SIZE passes of `bench.py --no-sft --no-sustain`): the decode step's dominant kernel
(gemv_kernel<1, 4>: RMSNorm + gate/up GEMV + SiLU*mul) and the prefill's (the fused gate/up GEMM).  Units and the gfx950 correction as
MI355X_MICROARCH.md's HBM section prescribes: the counters are KiB per launch; FETCH_SIZE counts the 128-B requests of wide coalesced 16-B/lane
streams as 64 B -> x2; WRITE_SIZE as reported.      usage: python tools/pmc_traffic_json.py <counters.txt> <out.json> <round tag>"""
import json
import re
import sys

src, out, tag = sys.argv[1], sys.argv[2], sys.argv[3]
rows = {}
for line in open(src):
    m = re.match(r"^(.*?)\s+(FETCH_SIZE|WRITE_SIZE)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)", line)
    if m:
        rows[(m.group(1).strip(), m.group(2))] = (int(m.group(3)), float(m.group(4)))


def pick(prefix):
    f = next((v for (k, c), v in rows.items() if k.startswith(prefix) and c == "FETCH_SIZE"), None)
    w = next((v for (k, c), v in rows.items() if k.startswith(prefix) and c == "WRITE_SIZE"), None)
    return f, w


H, F, S = 3584, 18944, 769
dec_alg = 2 * F * H * 2 + H * 2 * 2 + F * 2
f, w = pick("void gemv_kernel<1, 4>")
doc = {"kernel": "gemv_kernel<1,4>",
       "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python bench.py --no-cpu-baseline --no-sft --no-sustain --steps 8 --warmup 2 "
                 f"(round {tag}, tools/pmc.sh via tools/validate_gpu.sh; per-kernel averages in profiles/{tag}_pmc_hbm_counters.txt)",
       "correction": "gfx950: FETCH_SIZE counts 128-B requests as 64 B for wide coalesced 16-B/lane streams -> x2 (MI355X_MICROARCH.md HBM section); WRITE_SIZE uncalibrated, taken as reported"}
if f and w:
    doc.update({"FETCH_SIZE_avg_KiB": f[1], "WRITE_SIZE_avg_KiB": w[1], "launches": f[0],
                "traffic_bytes_per_launch": int(f[1] * 1024 * 2 + w[1] * 1024), "algorithmic_bytes_per_launch": dec_alg})
f, w = pick("void gemm256_kernel<2, 0, false, false, 7, 256, true>")
if f and w:
    alg = (S * H + 2 * F * H + S * F) * 2                   # A once + both weight matrices once + the bf16 output
    t = int(f[1] * 1024 * 2 + w[1] * 1024)
    doc["prefill_gateup"] = {"kernel": "gemm256_kernel<2,0,false,false,7,256,true> (fused gate/up GEMM of the prefill, S = 769, leftover row in the last row tile)",
                             "FETCH_SIZE_avg_KiB": f[1], "WRITE_SIZE_avg_KiB": w[1], "traffic_bytes_per_launch": t, "algorithmic_bytes_per_launch": alg,
                             "ratio": round(t / alg, 3), "source": f"same rocprofv3 --pmc passes as the decode kernel (profiles/{tag}_pmc_hbm_counters.txt), FETCH x2 on gfx950; L2 fills"}
json.dump(doc, open(out, "w"), indent=1)
print(json.dumps(doc)[:400])
