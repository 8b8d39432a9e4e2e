cd /root/repo
g++ -std=c++17 -O1 -g -pthread tests/cpp/test_two_devices.cpp -o /tmp/t2d -Lkzero_amd -lkzhip -Wl,-rpath,$PWD/kzero_amd
python3 -c "
from kzero_amd import synth
open('/tmp/chess_2x256.kzm','wb').write(synth.random_model('chess', 2, 256, 'attention', seed=5))"
for cfg in "/tmp/chess_2x256.kzm f16" "/tmp/chess_2x256.kzm f32split16" "tests/golden/go9_2x16_conv.kzm f32" "tests/golden/ataxx7_4x64.kzm f16"; do
  set -- $cfg
  echo "== $cfg"
  KZ_TWO_DEVICES_REHEARSE=1 timeout 60 /tmp/t2d $1 $2; echo "rc=$?"
done
which gdb
