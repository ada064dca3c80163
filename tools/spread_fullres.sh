# NOTE: these knobs are constants in the product build: build with CRD_EXTRA_FLAGS=-DCRD_DEV_SWITCHES python -m camradepth_amd.build
# and run with CRD_DEV_SWITCHES=1 (camradepth_amd/csrc/common.h: crd_dev_int; camradepth_amd/engine.py: _dev_int)
# spread of the 928x1600 golden comparison over launch geometries that only regroup float partial sums
for tw in 16 32; do for gs in 128 256; do
  echo "CRD_DW_TW=$tw CRD_GN_SMALL=$gs"
  CRD_DW_TW=$tw CRD_GN_SMALL=$gs timeout 300 python -m pytest tests/test_gpu_model.py -q -s -k "928x1600 or rmse_gap" 2>&1 | grep -E "vs reference|vs the reference|passed|failed"
done; done
