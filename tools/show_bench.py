import json,sys
d=json.load(open(sys.argv[1]))
print("value %.0f qps  %.1f us/step  roofline %s" % (d["value"], d["ms_per_step"]*1e3, {k:(round(v,3) if isinstance(v,float) else v) for k,v in d["roofline"].items()}))
for k in ("north_star_T","config_K2_head","cpu_baseline"):
    if k in d: print(k, {a:(round(b,4) if isinstance(b,float) else b) for a,b in d[k].items()})
