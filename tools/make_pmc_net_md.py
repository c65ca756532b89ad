"""profiles/archive/r03_pmc_net.md from the two raw SQ passes (tools/archive/refresh_profiles_r03.sh): python tools/make_pmc_net_md.py profiles/r03"""
import re
import sys

pre = sys.argv[1]
val = {}
for k in (1, 2):
    for line in open('%s_pmc_net_select_%d.md' % (pre, k)):
        m = re.match(r'\| (.*?) \| (SQ\w+) \| (\d+) \| ([\d.]+) \| ([\d.]+) \|', line)
        if m:
            kern = 'net' if 'k_v80_net_h2' in m.group(1) else 'select' if 'k_select' in m.group(1) else None
            if kern:
                val[(kern, m.group(2))] = float(m.group(4))
g = lambda k, c: val.get((k, c), 0.0)  # noqa: E731
W_NET, W_SEL = 256 * 12, 4096           # waves per dispatch
rows = [('kernel duration, cycles (SQ_BUSY_CYCLES)', '%.0f' % g('net', 'SQ_BUSY_CYCLES'), '%.0f' % g('select', 'SQ_BUSY_CYCLES')),
        ('MFMA instructions per SIMD (SQ_INSTS_MFMA / 32)', '%.0f' % (g('net', 'SQ_INSTS_MFMA') / 32), '-'),
        ('**MFMA busy cycles per SIMD (SQ_VALU_MFMA_BUSY_CYCLES / 32) = MFMA utilisation**',
         '%.0f = **%.1f %%** of the kernel' % (g('net', 'SQ_VALU_MFMA_BUSY_CYCLES') / 32, 100 * g('net', 'SQ_VALU_MFMA_BUSY_CYCLES') / 32 / max(1, g('net', 'SQ_BUSY_CYCLES'))), '0'),
        ('MFMA ops (SQ_INSTS_VALU_MFMA_MOPS_F16 x 512 flops, all 32 slices)',
         '%.2f GFLOP executed per forward (3 MFMAs per algorithmic product + padding; algorithmic 4.27 GFLOP)' % (g('net', 'SQ_INSTS_VALU_MFMA_MOPS_F16') * 32 * 512 / 1e9), '-')]
for c, label in (('SQ_WAIT_ANY', 'parked at s_waitcnt / barrier'), ('SQ_WAIT_INST_ANY', 'issue stalls'), ('SQ_ACTIVE_INST_ANY', 'issuing'),
                 ('SQ_ACTIVE_INST_VALU', '... VALU'), ('SQ_ACTIVE_INST_SCA', '... scalar')):
    rows.append(('%s, share of SQ_WAVE_CYCLES (%s)' % (c, label),
                 '%.1f %%' % (100 * g('net', c) / max(1, g('net', 'SQ_WAVE_CYCLES'))), '%.1f %%' % (100 * g('select', c) / max(1, g('select', 'SQ_WAVE_CYCLES')))))
rows.append(('VALU instructions per wave', '%.0f' % (g('net', 'SQ_INSTS_VALU') * 32 / W_NET), '%.0f' % (g('select', 'SQ_INSTS_VALU') * 32 / W_SEL)))
rows.append(('SALU instructions per wave', '%.0f' % (g('net', 'SQ_INSTS_SALU') * 32 / W_NET), '%.0f' % (g('select', 'SQ_INSTS_SALU') * 32 / W_SEL)))
rows.append(('LDS bank-conflict cycles / LDS active cycles per CU', '%.0f / %.0f' % (g('net', 'SQ_LDS_BANK_CONFLICT') / 8, g('net', 'SQ_ACTIVE_INST_LDS') / 8 * 4),
             '%.0f / %.0f' % (g('select', 'SQ_LDS_BANK_CONFLICT') / 8, g('select', 'SQ_ACTIVE_INST_LDS') / 8 * 4)))
print('# PMC summary of the two kernels of a round (rocprofv3 --pmc, two SQ passes of `bench.py --steps 1 --warmup 1 --preroll-plies 0`; raw: %s_pmc_net_select_[12].md)\n' % pre.split('/')[-1])
print('Values are averages per dispatch and per counter slice (32 slices: one per 8 CUs = 32 SIMDs); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x4 = cycles).\n')
print('| quantity | k_v80_net_h2 | k_select |\n|---|---|---|')
for r in rows:
    print('| %s | %s | %s |' % r)
