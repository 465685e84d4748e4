for env in "X=1" "MLP_HYPER_HEAVY=1048576" "MLP_HYPER_HEAVY=2097152" "MLP_HYPER_HEAVY=4194304" "MLP_RATIO_ONE=0" "MLP_HYPER_BACKOFF=2" "MLP_HYPER_BACKOFF=4" "MLP_GRAPH_ITERS=1"; do
  echo "== $env"; env $env python tools/hyper_profile.py 2>&1 | grep -v Warn | grep "MLP_HYPER=1:" | cut -c1-260
done
