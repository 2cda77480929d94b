# Seed 60016001's sequence around configurations #670 .. #684 under HSA_ENABLE_SDMA=0, again and again (SOAK_SKIP: the configurations before it only draw their random
# numbers), alternating the library's bounce-buffer copies (default) with the direct pageable hipMemcpyAsync of rounds 1-5 (COMET_COPY_DIRECT=1) on the SAME box.
# usage: r6_hunt.sh <pairs> <skip> <seconds>
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export COMET_GUARDS=1
N=${1:-14}; SKIP=${2:-600}; SEC=${3:-22}
for i in $(seq 1 $N); do
  ( HSA_ENABLE_SDMA=0 SOAK_SKIP=$SKIP timeout 200 python tools/soak.py $SEC 60016001 flat,ivf,pq,ivfpq 2>&1 | tail -1 ) > gpurun_out/r6_hunt4_bounce_$i.log 2>&1
  ( COMET_COPY_DIRECT=1 HSA_ENABLE_SDMA=0 SOAK_SKIP=$SKIP timeout 200 python tools/soak.py $SEC 60016001 flat,ivf,pq,ivfpq 2>&1 | tail -1 ) > gpurun_out/r6_hunt4_direct_$i.log 2>&1
done
echo "bounce (default): $(grep -l 'soak OK' gpurun_out/r6_hunt4_bounce_*.log | wc -l) clean, $(grep -l MISMATCH gpurun_out/r6_hunt4_bounce_*.log | wc -l) mismatches"
echo "direct (COMET_COPY_DIRECT=1): $(grep -l 'soak OK' gpurun_out/r6_hunt4_direct_*.log | wc -l) clean, $(grep -l MISMATCH gpurun_out/r6_hunt4_direct_*.log | wc -l) mismatches"
grep -h "MISMATCH" gpurun_out/r6_hunt4_*.log | cut -c1-900
