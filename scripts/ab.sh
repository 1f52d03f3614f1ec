# same-box A/B of library builds: bash scripts/ab.sh <rounds> <variant.so ...>   (variants under ab/, copied over the package's library in turn)
R=${GRAFT_REPO_ROOT:-.}
N=$1; shift
for i in $(seq $N); do
  for v in "$@"; do
    cp $R/ab/$v.so $R/genomic-resistance-mapping-grm-_amd/libgrmkmer.so
    echo -n "$v "; bash $R/scripts/head_quick.sh | cut -c1-260
  done
done
