# Timing ablations of conv_bf16_blk_kernel (results wrong by construction): -DRCA_BLK_ABL=1 no conv_in arithmetic in the fused layer,
# =2 no MFMAs, =3 no output stores.  usage on the GPU box: bash scripts/bf16_abl.sh <tag> <flag value or 0> [library]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1
if [ "$2" != "0" ]; then export RCA_EXTRA_HIPCC_FLAGS=-DRCA_BLK_ABL=$2; export RCA_LIB_PATH=$3; fi
rm -rf /tmp/bfabl_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bfabl_$TAG -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-duplex --no-trim-leg --no-cli-leg > /dev/null 2>&1
python3 $R/scripts/kstats.py /tmp/bfabl_$TAG 40 | grep "blk_kernel" | sed "s/^/$TAG /" | cut -c1-62,97-125
