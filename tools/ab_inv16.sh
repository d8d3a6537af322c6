for rep in 1 2 3; do
for lib in "" tools/libparsy_inv16old.bin; do
  for w in nd24k ex15 mid3d; do
    echo -n "lib=${lib:-product} $w: "; PARSY_LIB=$lib python tools/one_factor.py $w 30 2>&1 | tail -1
  done
done; done
echo -n "flan product: "; python tools/one_factor.py flan 4 2>&1 | tail -1
echo -n "flan old: "; PARSY_LIB=tools/libparsy_inv16old.bin python tools/one_factor.py flan 4 2>&1 | tail -1
