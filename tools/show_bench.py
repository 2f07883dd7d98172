import json,sys
j=json.load(open(sys.argv[1]))
print("ms", j["ms_per_step"], "train", j.get("train_step_graph_ms"), j.get("train_step_error"))
r=j.get("roofline_mfma")
if r:
    print(r["discriminator_gemm_ms_per_step"], r["discriminator_gemm_tflops_overall"], r["kernel"], r["frac"])
    for k,v in list(r["instances"].items())[:12]: print(f"{v['avg_launch_us']*v['launches_per_step']:8.0f} us  {v['launches_per_step']:.0f} x {v['avg_launch_us']:7.1f}  {v['frac_of_dense_peak']:.3f}  {k}")
