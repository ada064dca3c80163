"""gpurun_out/pmc_encoder/summary.txt (tools/pmc_encoder.sh) -> profiles/rNN_pmc_kernels.md: per kernel family the raw counters and the
derived shares the north star asks for (MFMA busy against the matrix pipes' cycles, wave time split into issuing / waiting,
HBM bytes against the kernel's active time).  Usage: python tools/pmc_encoder_md.py <summary.txt> <out.md>"""
import collections
import re
import sys

src, out = sys.argv[1], sys.argv[2]
runs, cur = collections.OrderedDict(), None
for line in open(src):
    if line.startswith("=="):
        cur = line[2:].strip()
        runs[cur] = collections.defaultdict(dict)
        continue
    m = re.match(r"\s+(.+?)\s{2,}(\S+)\s+(?:launches=\s*(\d+)\s+avg=\s*([\d.e+]+)\s+)?sum=\s*([\d.e+]+)", line)
    if m and cur:
        k, c, n, avg, tot = m.groups()
        runs[cur][k.strip()][c] = (int(n) if n else 0, float(tot))
SIMDS, XCDS = 1024, 8
with open(out, "w") as f:
    f.write("# Hardware counters of the encoder kernels\n\n"
            "`tools/pmc_encoder.sh` on one MI355X: one `rocprofv3 --pmc` pass per counter set (no tracing), program = `python3 tools/run_forward.py`\n"
            "(eager iterations of the benchmark configuration: base, 8 x 7 x 256 x 416).  Raw sums are per RUN (3 iterations); launches = launches in the run.\n\n"
            "Derived columns: **wave time** = SQ_WAVE_CYCLES split into ACTIVE (issuing: SQ_ACTIVE_INST_ANY), WAIT (parked on s_waitcnt / barrier:\n"
            "SQ_WAIT_ANY) and ISSUE-STALL (SQ_WAIT_INST_ANY), as shares of the three's sum; **MFMA busy** = SQ_VALU_MFMA_BUSY_CYCLES / (1024 matrix pipes x\n"
            "GRBM_GUI_ACTIVE / 8 XCDs), i.e. the share of the chip's matrix-pipe cycles the kernel kept busy while it ran; **waves/launch** = SQ_WAVES /\n"
            "launches (a full wave of the chip is 256 CUs x 8-32 waves); **HBM** = (2 x FETCH_SIZE + WRITE_SIZE) KB per launch (gfx950: FETCH_SIZE counts\n"
            "half the bytes of wide reads, MI355X_MICROARCH.md) and, divided by the kernel's active time at 2.1 GHz, GB/s.\n\n")
    for run, ks in runs.items():
        f.write(f"## {run}\n\n| kernel family | launches | waves/launch | ACTIVE | WAIT | ISSUE-STALL | MFMA busy | VALU inst/wave | HBM MB/launch | HBM GB/s |\n|---|---|---|---|---|---|---|---|---|---|\n")
        for k, c in ks.items():
            if "SQ_WAVES" not in c:
                continue
            n = c["SQ_WAVES"][0] or 1
            waves = c["SQ_WAVES"][1]
            act, wait, stall = (c.get(x, (0, 0))[1] for x in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"))
            tot = act + wait + stall or 1
            gui = c.get("GRBM_GUI_ACTIVE", (0, 0))[1]
            mfma = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0))[1]
            mfma_share = mfma / (SIMDS * gui / XCDS) if gui else 0
            valu = c.get("SQ_INSTS_VALU", (0, 0))[1] / waves if waves else 0
            kb = 2 * c.get("FETCH_SIZE", (0, 0))[1] + c.get("WRITE_SIZE", (0, 0))[1]
            secs = gui / XCDS / 2.1e9
            label = k if n else k
            f.write(f"| `{label}` | {n if n else '-'} | {waves / n:.0f} | {act / tot:.0%} | {wait / tot:.0%} | {stall / tot:.0%} | {mfma_share:.1%} | {valu:.0f} | "
                    f"{kb / n / 1e3:.1f} | {kb * 1e3 / secs / 1e9 if secs else 0:.0f} |\n")
        f.write("\n")
    f.write("Raw counter sums: `gpurun_out/pmc_encoder/summary.txt` of the run (copied below).\n\n```\n" + open(src).read() + "```\n")
print("wrote", out)
