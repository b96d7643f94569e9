import sys, json
d=json.loads(sys.stdin.read()); k=d["kernels"]
print(sys.argv[1], d["value"], {x:(round(k[x]["avg_us"]),k[x]["launches"]) for x in k}, d.get("roofline",{}) and {a:d["roofline"].get(a) for a in ("avg_launch_us_serial",)})
