# factorisation times of the nested-dissection LU over the cases DESIGN.md quotes (development aid)
cd /root/repo
for c in "S30k" "S120k" "S500k" "C40k" "C160k" "C160k --complex" "C300k"; do
  echo "== $c"; timeout -k 10 200 python tools/bench_ndlu.py --case $c --refactors 3 2>&1 | grep -E "refactor|residual|solve " | tail -3
done
