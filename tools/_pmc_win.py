import glob, sys, pandas as pd
for d in sys.argv[1:]:
    fs = glob.glob(d + "/**/*counter_collection.csv", recursive=True)
    df = pd.concat([pd.read_csv(f) for f in fs])
    df = df[df.Kernel_Name.str.contains("win_attn")]
    df["k"] = df.Kernel_Name.str.extract(r"(win_attn_\w\w\w)")
    g = df.groupby(["k", "Grid_Size", "Counter_Name"]).Counter_Value.mean().reset_index()
    g["MB"] = g.Counter_Value * 1024 / 1e6
    print(g.to_string())
