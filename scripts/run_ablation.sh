source scripts/bench_matrix.sh true
show() { python3 - <<PY
import json
ops=json.load(open("gpurun_out/ops_$1.json"))
print("   " + "  ".join("%s %.1f" % (o['name'].replace('backbone.','').replace('classifier.0','head'), o['ms']*1000) for o in ops if o['name'] in ('classifier.0','backbone.layer4.1.conv2','backbone.layer3.1.conv2')))
PY
}
for a in 0 1 2 3 4; do
NBC_CONV_ABLATE=$a NBC_CONV_MFMA32=1 run bf16_t5_abl$a --steps 20 --warmup 3 --streams 1 --conv-tile 5; show bf16_t5_abl$a
done
for a in 0 1 2 3 4; do
NBC_CONV_ABLATE=$a NBC_CONV_MFMA32=1 run bf16_b8t3_abl$a --steps 6 --warmup 2 --streams 1 --conv-tile 3 --batch 8; show bf16_b8t3_abl$a
done
