import json,sys
d=json.load(open(sys.argv[1]))
for o in d["other_configs"]:
    print(o["config"], "step", round(o["step_ms"],2), "build", round(o["build_ms"],2), "join", round(o["join_ms"],2), "d2h+host", round(o["d2h_and_host_ms"],2), "ok", o.get("sum_shared_equals_holder_pairs"))
