import sys, json
for path in sys.argv[1:]:
    d = json.loads([x for x in open(path) if x.startswith("{")][-1])
    r = d["roofline"]
    fam = r["families"]
    print(path.split("/")[-1], "step", d["ms_per_step"], "median", d["step_ms"]["median"], "serial", r["step"]["serial_step_ms"])
    print("   " + " | ".join(f"{k.replace('_kernel','')} {v['ms_per_step']}" for k, v in list(fam.items())[:12]))
