cd $GRAFT_REPO_ROOT
cp gan-segmentation_amd/csrc/libgsa_hip.so /tmp/cur.so
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bf16.py -m gpu -x -q > gpurun_out/r2_t45.log 2>&1; tail -2 gpurun_out/r2_t45.log
for rep in 1 2; do for v in old new; do cp gan-segmentation_amd/csrc/ab/$v.so gan-segmentation_amd/csrc/libgsa_hip.so; echo "== $v"; timeout -k 10 120 python bench.py --batch 8 --steps 30 --no-secondary --no-cpu-baseline --layers 2> gpurun_out/ab_$v$rep.txt | grep -o '"value": [0-9.]*\|matches_oracle": [a-z]*' | tr '\n' ' '; echo; done; done
for v in old new; do cp gan-segmentation_amd/csrc/ab/$v.so gan-segmentation_amd/csrc/libgsa_hip.so; timeout -k 10 120 python bench.py --gan cars --batch 4 --precision bf16 --steps 30 --no-secondary --no-cpu-baseline 2>/dev/null | grep -o '"value": [0-9.]*'; done
cp /tmp/cur.so gan-segmentation_amd/csrc/libgsa_hip.so
for v in old1 new1; do echo == $v; grep -E "subpixel" gpurun_out/ab_$v.txt | sed 's/void gsa:://; s/(gsa::ConvParams)//' | awk -F'|' '{print $2}' | cut -c1-40 | tr '\n' ';'; echo; done
