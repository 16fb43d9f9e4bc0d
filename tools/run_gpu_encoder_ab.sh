cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "0 0 1" "1 0 1" "1 1 1" "1 1 0" "0 0 0"; do
  set -- $cfg
  echo "defer $1 inplace $2 gram $3 rep $rep"
  SAT_DEFER_BN3=$1 SAT_DEFER_INPLACE=$2 SAT_GRAM_BN3=$3 timeout -k 10 200 python tools/encoder_only.py 60 2>/dev/null | tail -2
done
done
