import json,sys
l=json.load(open(sys.argv[1]))
print("ms/step",round(l["ms_per_step"],4),"host",round(l["host_issue_ms_per_step"],4),"value %.3e"%l["value"])
for k,v in l["kernels"].items(): print("  ",k, round(v["avg_ms"],4), v["launches"])
