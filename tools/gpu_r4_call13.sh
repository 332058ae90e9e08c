#!/bin/bash
# round 4, call 13: APE on the EVA-01 MIM ViT-g (vit_eva.py) at FULL size vs the reference fixture (fp32 + both 16-bit flavours), G_A bench line
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
O=gpurun_out/call13
mkdir -p $O
APE_WRITE_PINS=$O APE_TEST_ALL_F16=1 timeout 900 python -m pytest tests/test_model_gpu.py -q -m gpu -s -k "G_A_1536" 2>&1 | grep -v Warning > $O/pytest.log; tail -4 $O/pytest.log | cut -c1-300
grep -n "G_A_1536\]" $O/pytest.log | cut -c1-400 | head -40

