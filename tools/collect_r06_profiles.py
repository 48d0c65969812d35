"""Copies the round's judged summaries from gpurun_out/ (scratch) into profiles/ (tracked) after tools/gpu_r06_final.sh, and condenses the
two PMC passes of tools/gpu_r06_pmc.sh into profiles/r06_pmc_kernels.txt.    python tools/collect_r06_profiles.py"""
import os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G, P = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
COPY = [("r6_bench.json", "r06_bench.json", "last_line"), ("r6_bench_detail.json", "r06_bench_detail.json", None),
        ("r6_bench_driver.json", "r06_bench_driver_style.json", "last_line"), ("r6_stress_linear.txt", "r06_stress_linear.txt", None),
        ("r6_pytest.log", "r06_pytest_gpu.log", None),
        ("profb_r06/r06_bench_rocprofv3_kernel_stats.csv", "r06_bench_rocprofv3_kernel_stats.csv", None),
        ("profb_r06/r06_pmc_traffic.json", "r06_pmc_traffic.json", None),
        ("profb16_r06/r06_bench16_rocprofv3_kernel_stats.csv", "r06_bench16_rocprofv3_kernel_stats.csv", None),
        ("profb16_r06/r06_pmc_traffic16.json", "r06_pmc_traffic16.json", None)]
for src, dst, how in COPY:
    s = os.path.join(G, src)
    if not os.path.exists(s):
        print("missing", src); continue
    if how == "last_line":
        open(os.path.join(P, dst), "w").write(open(s).read().strip().splitlines()[-1] + "\n")
    else:
        shutil.copyfile(s, os.path.join(P, dst))


def sums(tag, kernel):
    txt = open(os.path.join(G, "prof_" + tag, "summary.txt")).read()
    stats = next((l for l in txt.splitlines() if kernel in l and l.startswith('"')), "")
    body = txt[txt.index("== PMC sums"):]
    blk = re.search(r"^([^\n]*" + kernel + r"[^\n]*)\n((?:    \S+ = \d+\n?)+)", body, re.M)
    c = {m.group(1): float(m.group(2)) for m in re.finditer(r"    (\S+) = (\d+)", blk.group(2))}
    return stats, blk.group(1), c


rows, blocks = [], []
for tag, kernel, frames in (("r06lin", "logo_eval_linear_kernel", 4096), ("r06scan", "logo_eval_pair_kernel", 4096)):
    stats, name, c = sums(tag, kernel)
    ms = float(stats.split(",")[3]) / 1e6 if stats else float("nan")          # (name, calls, total ns, average ns, ...)
    stats = ",".join(stats.split(",")[:4])
    wc = c["SQ_WAVE_CYCLES"]
    rows.append("#   %-26s %-8.3f %-26.2f %-14.2f %-18.2f %-13.2f %-23.2f %d / %d / %d / %d   %.3f   %d" % (
        kernel, ms, c["SQ_ACTIVE_INST_ANY"] / wc, c["SQ_ACTIVE_INST_VALU"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_WAIT_ANY"] / wc,
        c["SQ_LDS_BANK_CONFLICT"] / max(1.0, c["SQ_LDS_IDX_ACTIVE"]), c["SQ_INSTS_VALU"] / frames, c["SQ_INSTS_VMEM_RD"] / frames,
        c["SQ_INSTS_LDS"] / frames, c["SQ_INSTS_SALU"] / frames, c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), c["SQ_WAVES"]))
    blocks.append("==== %s: %s\n  stats: %s\n%s\n%s" % (tag, kernel, stats, name, "".join("    %s = %d\n" % (k, v) for k, v in sorted(c.items()))))
with open(os.path.join(P, "r06_pmc_kernels.txt"), "w") as f:
    f.write("# rocprofv3 --pmc counter sums at HEAD of round 6 (tools/gpu_r06_pmc.sh -> tools/gpu_prof.sh: one counter set per pass, never combined with\n"
            "# tracing; 4096 frames of 1440x1080 8-bit, one launch each).  SQ_* cycle counters are in units of 4 clocks, summed over all waves / SIMDs.\n"
            "# Round 5's figures for the same launches: profiles/r05_pmc_kernels.txt.\n"
            "#   kernel                     ms/4096  issuing (ACTIVE_INST_ANY)  of which VALU  ready, not issued  in s_waitcnt  LDS bank-conflict share  "
            "VALU / VMEM-read / LDS / SALU wave-instructions per frame   L2 hit rate  waves\n")
    f.write("\n".join(rows) + "\n" + "\n".join(blocks))
print(open(os.path.join(P, "r06_pmc_kernels.txt")).read()[:1500])
