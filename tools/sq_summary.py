"""Per-wave issue statistics of the step kernels from a rocprofv3 --pmc SQ_* pass:  python tools/sq_summary.py gpurun_out/<tag>/pmc_sq"""
import collections, csv, glob, sys
for fn in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list)); meta = {}
    for r in csv.DictReader(open(fn)):
        k = r['Kernel_Name']
        if 'step_kernel' not in k:
            continue
        acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
        meta[k] = dict(vgpr=r['VGPR_Count'], agpr=r['Accum_VGPR_Count'], lds=r['LDS_Block_Size'], scratch=r['Scratch_Size'], sgpr=r['SGPR_Count'], grid=r['Grid_Size'])
    for k, v in acc.items():
        m = {c: sum(x) / len(x) for c, x in v.items()}
        print(k, meta[k])
        if 'SQ_WAVES' in m:
            w = m['SQ_WAVES']
            tot = m['SQ_INSTS_VALU'] + m['SQ_INSTS_SALU'] + m['SQ_INSTS_LDS']
            print('  per wave: VALU %.0f SALU %.0f LDS %.0f total %.0f | slots (wave quad-cycles) %.0f | wait_any %.0f | active_inst_any %.0f | issue frac %.3f'
                  % (m['SQ_INSTS_VALU'] / w, m['SQ_INSTS_SALU'] / w, m['SQ_INSTS_LDS'] / w, tot / w, m['SQ_WAVE_CYCLES'] / w, m['SQ_WAIT_ANY'] / w,
                     m.get('SQ_ACTIVE_INST_ANY', 0) / w, tot / m['SQ_WAVE_CYCLES']))
        else:
            print('  ', m)
