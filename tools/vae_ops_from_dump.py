import json,sys
d=json.load(open(sys.argv[1]))
rows=[r for r in d["musetalk_rows"] if r["layer"].startswith("vae:")]
for r in rows:
    if r["ms"]>0.04: print(f"{r['layer'][:58]:58s} {r['kernel'][:60]:60s} {r['ms']*1e3:7.1f} us")
print("vae total", sum(r["ms"] for r in rows))
