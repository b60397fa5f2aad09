# same-box A/B of builds of the library on the PnP call rates: bash scripts/gpu_pnp_ab.sh <rounds> <libA> <libB> [<libC> ...]
# (box-to-box spread is +-1 %: only alternating runs on ONE box are trusted for differences of a few per cent)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=$1; shift
mkdir -p gpurun_out/r04
for i in $(seq $R); do
  for L in "$@"; do
    echo -n "$(basename $L): "; CHIP_ALLOW_LIB_OVERRIDE=1 CHIP_LIB=$L timeout 300 python scripts/gpu_pnp_rates.py 2>&1 | tail -1
  done
done | tee gpurun_out/r04/pnp_ab.txt
