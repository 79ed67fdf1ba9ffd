#!/usr/bin/env python3
"""Derived per-kernel table from the raw counter averages tools/pmc_kernel.sh collects (tools/pmc_dump.py lines):
kernel cycles = SQ_BUSY_CYCLES / 32 shader engines; matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / (cycles x 1024 SIMDs);
LDS = SQ_LDS_IDX_ACTIVE / (cycles x 256 CUs); wavefront time split = SQ_WAIT_ANY | SQ_WAIT_INST_ANY | SQ_ACTIVE_INST_ANY, each
/ SQ_WAVE_CYCLES; L2 hit = TCC_HIT / (HIT + MISS).   usage: pmc_table.py <raw.txt> > table.txt"""
import collections
import re
import sys

vals = collections.defaultdict(dict)
count = {}
for line in open(sys.argv[1]):
    m = re.match(r"imf::(.*?)\s+(\S+)\s+n=\s*(\d+)\s+avg=\s*([\d.]+)\s*$", line)
    if m:
        vals[m.group(1)][m.group(2)] = float(m.group(4))
        count[m.group(1)] = int(m.group(3))
print("%-44s %5s %9s %8s %7s %6s %6s %6s %6s %6s %8s  insts per launch: mfma / valu / lds / vmem_rd" %
      ("kernel", "n", "cycles", "us@2.1G", "MFMA%", "LDS%", "park%", "stall%", "act%", "L2hit%", "bankcf"))
for k in sorted(vals):
    v = vals[k]
    g = lambda name: v.get(name, float("nan"))
    cyc = g("SQ_BUSY_CYCLES") / 32.0
    wave = g("SQ_WAVE_CYCLES")
    hit, miss = g("TCC_HIT_sum"), g("TCC_MISS_sum")
    print("%-44s %5d %9.0f %8.1f %7.1f %6.1f %6.1f %6.1f %6.1f %6.1f %8.0f  %.0f / %.0f / %.0f / %.0f" % (
        k, count[k], cyc, cyc / 2100.0, 100 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024), 100 * g("SQ_LDS_IDX_ACTIVE") / (cyc * 256),
        100 * g("SQ_WAIT_ANY") / wave, 100 * g("SQ_WAIT_INST_ANY") / wave, 100 * g("SQ_ACTIVE_INST_ANY") / wave,
        100 * hit / (hit + miss) if hit + miss == hit + miss and hit + miss > 0 else float("nan"), g("SQ_LDS_BANK_CONFLICT"),
        g("SQ_INSTS_MFMA"), g("SQ_INSTS_VALU"), g("SQ_INSTS_LDS"), g("SQ_INSTS_VMEM_RD")))
