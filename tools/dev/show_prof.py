import pstats, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*.prof"))):
    print("=" * 20, f)
    st = pstats.Stats(f)
    st.sort_stats("cumulative").print_stats(28)
