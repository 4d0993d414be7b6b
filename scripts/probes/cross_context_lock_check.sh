#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
T=$(mktemp -d); cp -r $R/include $T/include
python - <<PY
p="$T/include/lvk/LiveVisionKit.hpp"; s=open(p).read()
old="""        give_back = !m_Overlap && input.context() != m_Ctx;
        }"""
new="""        give_back = false; if (!m_Overlap && input.context() != m_Ctx) input.context()->wait_for(*m_Ctx);      // (the pre-round-5 code: the wait inside the lock)
        }"""
assert old in s; open(p,"w").write(s.replace(old,new,1))
PY
TL=$(python -c "import torch,os; print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
for inc in $R/include $T/include; do
  g++ -std=c++20 -O1 -pthread -DRUN_ON_GPU -I$inc -o $T/conf $R/tests/cpp/plugin_conformance.cpp -L$R/livevisionkit_amd -llvk_hip -L$TL -l:libamdhip64.so -Wl,-rpath,$R/livevisionkit_amd -Wl,-rpath,$TL
  echo "== include = $inc"; timeout 200 $T/conf --threads-and-files $T 2>&1 | grep -i "cross-context\|threads ok\|file input" ; echo "rc=$?"
done
