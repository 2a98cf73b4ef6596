"""Summarise a rocprofv3 rocpd database: per kernel (name, grid) launches, average and total time."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6, d.grid_size_x, d.grid_size_y "
     f"from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x, d.grid_size_y order by 4 desc")
print("kernel,launches,avg_us,total_ms,grid_x,grid_y")
for r in list(db.execute(q))[: int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"{r[0][:70]},{r[1]},{r[2]:.1f},{r[3]:.2f},{r[4]},{r[5]}")
