import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"]/1e9, d["ms_per_step"], d["roofline"]["kernels"], d["roofline"]["traffic_note"])
print(json.dumps(d.get("other_configs"), indent=0)[:3500])
print(json.dumps(d["e2e"].get("multi_sample"), indent=0)[:3000], flush=True)
print({k: (v.get("wall_s"), v.get("reads_per_s")) for k,v in d["e2e"].items() if isinstance(v, dict) and "wall_s" in v})
