"""FETCH_SIZE / WRITE_SIZE of the dominant conv launch (tools/run_pmc.sh fetch write passes) -> profiles/rNN_pmc_traffic.json
    python tools/pmc_traffic.py gpurun_out/pmcN out.json
FETCH_SIZE / WRITE_SIZE are in KiB; the read side is doubled as MI355X_MICROARCH.md prescribes for wide (16 B / lane)
coalesced reads on gfx950."""
import csv, glob, json, sys
KERNEL = sys.argv[3] if len(sys.argv) > 3 else "k_conv_gather"         # third argument: kernel name (k_conv_wide for the CLIP workload)
DESC = {"k_conv_gather": "k_conv_gather, L0 3^3 96->96 bf16 forward, 8-scene batch (tools/pmc_conv.py)",
        "k_conv_wide": "k_conv_wide, L0 3^3 512->512 bf16 forward, 8-scene batch (tools/pmc_conv_wide.py)"}[KERNEL]
vals = {"FETCH_SIZE": [], "WRITE_SIZE": []}
for f in sorted(glob.glob(sys.argv[1] + '/*/p_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        if KERNEL in r['Kernel_Name'] and r['Counter_Name'] in vals:
            vals[r['Counter_Name']].append(float(r['Counter_Value']))
fetch = sum(vals["FETCH_SIZE"]) / max(len(vals["FETCH_SIZE"]), 1) * 1024
write = sum(vals["WRITE_SIZE"]) / max(len(vals["WRITE_SIZE"]), 1) * 1024
out = {"kernel": DESC, "launches": len(vals["FETCH_SIZE"]),
       "fetch_bytes_raw": fetch, "fetch_bytes_corrected": 2 * fetch, "write_bytes": write, "traffic_bytes": 2 * fetch + write,
       "note": "rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes; read side x2 (gfx950 wide-read under-count)"}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(out)
