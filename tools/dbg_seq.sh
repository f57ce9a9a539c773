DUMP=/tmp/ref.npz timeout 200 python tools/dbg_xchg.py pushed:8:1 2>&1 | grep -v amdgpu.ids | cut -c1-300 | head -5
for first in "pushed:2:1,1,3,20" "pushed:2:1" "pushed:2:3" "pushed:2:1,1,3,20:keep"; do
echo "== $first"
CMP=/tmp/ref.npz timeout 200 python tools/dbg_xchg.py $first pushed:8:1 2>&1 | grep -v amdgpu.ids | cut -c1-300 | grep -v "nan in\|rank [1-7]\| M2 \| W " | head -12
done
echo "== NO_FOLD"
SMARTIES_HIP_NO_FOLD=1 CMP=/tmp/ref.npz timeout 200 python tools/dbg_xchg.py pushed:2:1,1,3,20 pushed:8:1 2>&1 | grep -v amdgpu.ids | cut -c1-300 | grep -v "nan in\|rank [1-7]\| M2 \| W " | head -12
