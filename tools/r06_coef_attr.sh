export RK_AB_LIB=qgs_amd/libqgs_hip_dev.so QGS_HIP_CACHE_DIR=/tmp/kc_ws; mkdir -p $QGS_HIP_CACHE_DIR
out=gpurun_out/r06_coef_attr.txt; : > $out
for v in "QGS_HIP_LDS_ASM_COEF=0" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_SKIP=1" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_SKIP=2" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_SKIP=4" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_SKIP=6" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_SKIP=7" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_LANES=4" "QGS_HIP_LDS_ASM_COEF=0 QGS_HIP_LDS_ASM_LANES=2" "QGS_HIP_LDS_ASM_COEF=0"; do
  echo "== $v" >> $out
  env $v timeout 600 python tools/lds228_time.py 2>&1 | grep -v amdgpu.ids | sed 's/{.*}//' >> $out
done
cat $out
