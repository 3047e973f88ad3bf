import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
import gpu_check as g
for n in [int(a) for a in sys.argv[1:]]:
    g.throughput("args/dog_slopes_mixed_args.txt", n, 20, True)
