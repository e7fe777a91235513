for i in 1 2; do
for E in "OCTA_NO_WGRAD8_FOLD=1" "" "OCTA_WG9_EPI=16 OCTA_WG8_EPI=3" "OCTA_WG9_EPI=8 OCTA_WG8_EPI=2 OCTA_WG9_MINSTEPS=16"; do
  env $E python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --launch graph 2>&1 | grep "timed region" | sed "s|^|[$E] |" | cut -c1-120
done; done
