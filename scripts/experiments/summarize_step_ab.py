import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l); print(f'{d["lib"]:22s}', *[f'{int(k)>>10}k:{v["ms_median"]*1e3:.1f}/{v["back_to_back_ms_median"]*1e3:.1f}' for k,v in d.items() if k!="lib"], [v["checksum"]%1000 for k,v in d.items() if k!="lib"])
