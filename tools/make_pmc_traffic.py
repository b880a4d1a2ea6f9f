"""Dev tool: turn the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE counter_collection.csv) of tools/gpu/profile.sh into
profiles/pmc_traffic.json -- per-kernel fabric-side bytes per launch -- stamped with the hash of the kernel sources it was
collected on (bench.py quotes `roofline.traffic` only for that build).
usage: python tools/make_pmc_traffic.py FETCH.csv WRITE.csv ROUND_TAG"""
import csv, json, sys, collections, os
sys.path.insert(0, '.')
from bench import kernel_source_hash

LABEL = [("k_obs<0, 1>", "k_obs<systematic>"), ("k_obs<0, 0>", "k_obs<stratified>"), ("k_obs<1, 1>", "k_obs<systematic>"), ("k_obs<1, 0>", "k_obs<stratified>"),
         ("k_step<", "k_step<trans+weight>"), ("k_lw_partials", "k_lw_partials"), ("k_resolve<0>", "k_resolve<W>"), ("k_resolve<1>", "k_resolve<P>"),
         ("k_local<0, true", "k_weights(normalize+local<W>)"), ("k_local<1, false, true", "k_local<P>(+resolve<W>)"), ("k_local<1, false, false", "k_local<P>"),
         ("k_apply<1, true, false", "k_apply<systematic>(+resolve<P>)"), ("k_apply<0, true, false", "k_apply<stratified>(+resolve<P>)"),
         ("k_apply<1, false, false", "k_apply<systematic>"), ("k_apply<0, false, false", "k_apply<stratified>")]


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = r["Kernel_Name"].replace("void bssm::", "").replace("bssm::", "")
        for pat, lab in LABEL:
            if name.startswith(pat) and int(r["Grid_Size"]) >= 256 * 512:       # the N = 2^20 launches only
                acc[lab].append(float(r["Counter_Value"]))
                break
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"round": sys.argv[3], "kernel_source_sha256": kernel_source_hash(),
       "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) of `python3 bench.py --steps 1 --warmup 0 --T 60 "
                 "--no-cpu-baseline --no-profile --no-pmmh --no-batch --no-configs`, N = 2^20; counter unit KiB; FETCH_SIZE doubled as "
                 "MI355X_MICROARCH.md (HBM section) prescribes for 16-B/lane streaming reads on gfx950; the working set (~36 MiB) is "
                 "Infinity-Cache resident, so these are fabric-side request bytes, not DRAM bytes",
       "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    out["kernels"][k] = {"fetch_size_KiB_raw": f, "write_size_KiB": w, "bytes_per_launch_corrected": (2 * f + w) * 1024}
json.dump(out, open(os.path.join("profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out["kernels"], indent=1))
