source scripts/bench_matrix.sh true
show() { python3 - <<PY
import json
ops=json.load(open("gpurun_out/ops_$1.json"))
print("   " + "  ".join("%s %.1f" % (o['name'].replace('backbone.','').replace('classifier.0','head'), o['ms']*1000) for o in ops if o['name'] in ('classifier.0','backbone.layer4.1.conv2','backbone.layer4.1.conv1','backbone.layer4.1.conv3','backbone.layer3.1.conv2','backbone.layer3.1.conv1')))
PY
}
for t in 5 3; do
run bf16_t${t}_m16 --steps 20 --warmup 3 --streams 1 --conv-tile $t; show bf16_t${t}_m16
NBC_CONV_MFMA32=1 run bf16_t${t}_m32 --steps 20 --warmup 3 --streams 1 --conv-tile $t; show bf16_t${t}_m32
done
run bf16_b8_t3_m16 --steps 6 --warmup 2 --streams 1 --conv-tile 3 --batch 8; show bf16_b8_t3_m16
NBC_CONV_MFMA32=1 run bf16_b8_t3_m32 --steps 6 --warmup 2 --streams 1 --conv-tile 3 --batch 8; show bf16_b8_t3_m32
run bf16_lat --steps 40 --warmup 5 --streams 1
run bf16_s4 --steps 80 --warmup 8 --streams 4
run bf16_b8 --steps 10 --warmup 2 --streams 1 --batch 8
