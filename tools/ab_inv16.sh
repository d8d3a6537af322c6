# same-box A/B of the walker changes of round 5: the product against tools/libparsy_inv16old.bin
# (tools/build_variant.sh inv16old -DPARSY_WALKER_LDS_TRSM -DPARSY_INVERT16_COLUMNS: the walker of round 4)
for rep in 1 2 3; do
for lib in "" tools/libparsy_inv16old.bin; do
  for w in nd24k ex15 mid3d 64x64x64:27; do
    echo -n "lib=${lib:-product} "; PARSY_LIB=$lib python tools/factor_median.py $w 40 2>&1 | tail -1
  done
done; done
echo -n "lib=product "; python tools/factor_median.py flan 6 2>&1 | tail -1
echo -n "lib=old "; PARSY_LIB=tools/libparsy_inv16old.bin python tools/factor_median.py flan 6 2>&1 | tail -1
