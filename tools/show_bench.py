import json,sys
d=json.load(open(sys.argv[1]))
print(d["value"], d["roofline"]["frac"], d["train_bs512_us_per_step"], d.get("bf16_train_rows_per_s"))
print(json.dumps(d["roofline_extra"]["train_bf16"])[:900])
print(json.dumps(d["other_configs"]["narrow_tables"])[:1200])
