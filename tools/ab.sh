# A/B of two builds of the library on the same box: smarties_amd/libsmarties_hip.so (A) and libsmarties_hip.so.B (B); $1 = command
cp smarties_amd/libsmarties_hip.so /tmp/A.so
for round in 1 2; do
  cp /tmp/A.so smarties_amd/libsmarties_hip.so; TAG=A timeout -k 5 200 bash -c "$1" 2>&1 | grep -v amdgpu.ids
  cp smarties_amd/libsmarties_hip.so.B smarties_amd/libsmarties_hip.so; TAG=B timeout -k 5 200 bash -c "$1" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/A.so smarties_amd/libsmarties_hip.so
