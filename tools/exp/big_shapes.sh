# one launch at a time, 2^20 .. 2^25 blocks: shapes of the large BC7 launch (one library per process)
set -u
export TMPDIR=/tmp
cd tools/exp
for n in 20 21 22 23 25; do
  L=256; [ $n -ge 23 ] && L=64; [ $n -ge 25 ] && L=24
  for lib in "$@"; do
    echo -n "2^$n  "
    python3 ab_multi.py --n $((1<<n)) --rounds 2 --launches $L --targets bc7 $lib | tr '\n' ' '
    echo
  done
done
