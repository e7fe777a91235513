for E in "OCTA_CONCURRENT_DISC=0" "OCTA_DISC_AFTER=forward" "OCTA_DISC_AFTER=decoder_3" "OCTA_DISC_AFTER=decoder_4" "OCTA_DISC_AFTER=encoder_4" "OCTA_DISC_AFTER=decoder_4"; do
  env $E python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-roofline --no-dice --sustained 0 --launch graph 2>&1 | grep "timed region" | sed "s|^|[$E] |" | cut -c1-120
done
