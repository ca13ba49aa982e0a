cd /root/repo
mkdir -p gpurun_out/r04d
{
echo "## default schedule, 2^26"; LOG2=26 STEPS=3 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so build_ab/r04_base.so
echo "## NT=196608 (3 waves/SIMD), 2^26"; ECL_HIP_MUL_NT=196608 LOG2=26 STEPS=3 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
echo "## NT=262144, 2^26"; ECL_HIP_MUL_NT=262144 LOG2=26 STEPS=3 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
echo "## uniform 2^20 pieces (R=8), 2^24"; ECL_HIP_MUL_FIRST=20 ECL_HIP_MUL_GROW=100 ECL_HIP_MUL_TOP=20 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so build_ab/r04_base.so
echo "## grow 150% from 2^18, 2^24"; ECL_HIP_MUL_GROW=150 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
echo "## grow 150% from 2^19 top 2^21, 2^24"; ECL_HIP_MUL_FIRST=19 ECL_HIP_MUL_GROW=150 ECL_HIP_MUL_TOP=21 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
echo "## uniform 2^21 pieces first 2^19, 2^24"; ECL_HIP_MUL_FIRST=19 ECL_HIP_MUL_GROW=400 ECL_HIP_MUL_TOP=21 bash tools/ab_mul.sh "22" build_ab/r04_mul_nommadd.so
} 2>&1 | tee gpurun_out/r04d/mulexp.txt
