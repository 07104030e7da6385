mkdir -p gpurun_out/r3
SGAN_TEST_VERBOSE=noassert python -m pytest -m gpu -q -s -k "step_vs_reference_golden or twostage_full or twostage_small or cgan_step or fcgan_step" tests/test_hip_step.py > gpurun_out/r3/verbose_steps.log 2>&1
grep -c "flips-clause" gpurun_out/r3/verbose_steps.log
SGAN_NO_PATCH_S2=1 SGAN_TEST_VERBOSE=noassert python -m pytest -m gpu -q -s -k "cgan_step_vs_reference_golden" tests/test_hip_step.py 2>&1 | grep "flips-clause" > gpurun_out/r3/verbose_cgan_nos2.log
cat gpurun_out/r3/verbose_cgan_nos2.log | head -20
