cd /root/repo
for rep in 1 2; do
for cfg in "1 1" "1 0" "0 1" "0 0"; do set -- $cfg
echo "== MFMA=$1 SORT=$2"; MDFRI_AX_MFMA=$1 SORT=$2 NB=8 python3 tools/pipeline_example.py 2>&1 | grep -i "stream\|GCN" | tail -3
done; done
python3 -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
