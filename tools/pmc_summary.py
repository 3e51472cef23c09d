"""Summarise tools/run_pmc.sh output: python tools/pmc_summary.py gpurun_out/pmcN"""
import collections, csv, glob, sys
res = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + '/*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'k_pack' in k or 'reduce' in k: continue
        if 'k_conv_halo' in k: kk = 'k_conv_halo (L0 3^3 96->96 bf16 forward, per launch)'
        elif 'k_conv_wide' in k: kk = 'k_conv_wide (2-D blocked wide-channel conv of the driver script, per launch)'
        elif 'k_conv_gather' in k: kk = 'k_conv_gather (dominant conv launch of the driver script, per launch)'
        elif 'k_wgrad_bf16' in k: kk = 'k_wgrad_bf16<3,3> (L0 3^3 96->96, per launch)'
        elif 'k_wgrad_ps' in k: kk = 'k_wgrad_ps<27,3> (L0 3^3 96->96, per launch)'
        else: continue
        res[kk][r['Counter_Name']].append(float(r['Counter_Value']))
for kk, d in res.items():
    print("\n==", kk)
    for c, v in sorted(d.items()):
        print("  %-36s launches=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
    if 'TCC_HIT_sum' in d:
        h, m = sum(d['TCC_HIT_sum']), sum(d['TCC_MISS_sum'])
        print("  -> L2 hit rate %.1f %%" % (100 * h / (h + m)))
    if 'TCP_TCC_READ_REQ_sum' in d:
        print("  -> L1 hit rate %.1f %%" % (100 * (1 - sum(d['TCP_TCC_READ_REQ_sum']) / sum(d['TCP_TOTAL_CACHE_ACCESSES_sum']))))
    if 'SQ_WAIT_ANY' in d:
        w = sum(d['SQ_WAVE_CYCLES'])
        print("  -> wave time: waiting %.0f %%, issuing %.0f %%, issue-stalled %.0f %%" % (
            100 * sum(d['SQ_WAIT_ANY']) / w, 100 * sum(d['SQ_ACTIVE_INST_ANY']) / w, 100 * sum(d['SQ_WAIT_INST_ANY']) / w))
    if 'FETCH_SIZE' in d:
        print("  -> HBM fetch %.3f GB (x2 if the gfx950 wide-read under-count applies)" % (sum(d['FETCH_SIZE']) / len(d['FETCH_SIZE']) * 1024 / 1e9))
