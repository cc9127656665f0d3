source scripts/bench_matrix.sh true
show() { python3 - <<PY
import json
ops=json.load(open("gpurun_out/ops_$1.json"))
print("   " + "  ".join("%s %.1f" % (o['name'].replace('backbone.','').replace('classifier.0','head'), o['ms']*1000) for o in ops if o['name'] in ('classifier.0','backbone.layer4.1.conv2','backbone.layer4.1.conv1','backbone.layer4.1.conv3','backbone.layer3.1.conv2','backbone.layer3.1.conv1','backbone.layer2.1.conv2','backbone.layer1.1.conv2')))
PY
}
for t in 5 14 2 15 6 16 0 17; do
run bf16_t${t} --steps 20 --warmup 3 --streams 1 --conv-tile $t; show bf16_t${t}
done
run bf16_lat --steps 40 --warmup 5 --streams 1
python3 -c "
import json; print(json.load(open('gpurun_out/bench_bf16_lat.json'))['config']['autotuned_tiles'])"
