source scripts/bench_matrix.sh true
show() { python3 - <<PY
import json
ops=json.load(open("gpurun_out/ops_$1.json"))
print("   " + "  ".join("%s %.1f" % (o['name'].replace('backbone.','').replace('classifier.0','head'), o['ms']*1000) for o in ops if o['name'] in ('classifier.0','backbone.layer4.1.conv2','backbone.layer4.1.conv1','backbone.layer4.1.conv3','backbone.layer3.1.conv2','backbone.layer3.1.conv1')))
PY
}
for t in 5 13 2 11; do
run bf16_t${t} --steps 20 --warmup 3 --streams 1 --conv-tile $t; show bf16_t${t}
done
run bf16_b8_t3 --steps 6 --warmup 2 --streams 1 --conv-tile 3 --batch 8; show bf16_b8_t3
run bf16_b8_t12 --steps 6 --warmup 2 --streams 1 --conv-tile 12 --batch 8; show bf16_b8_t12
