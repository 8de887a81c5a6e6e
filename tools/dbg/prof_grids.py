"""grid / workgroup sizes per kernel out of a rocprofv3 kernel trace (the `kernels` view of the sqlite file)"""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
print(cols)
want = [x for x in ("grid_x", "grid_size_x", "grid_size", "workgroup_x", "workgroup_size_x", "workgroup_size", "lds_size", "lds_block_size", "scratch_size") if x in cols]
q = "select name, count(*), avg(end-start)/1e3, %s from kernels group by name, %s order by 3 desc" % (", ".join(want), ", ".join(want))
for r in c.execute(q):
    print("%-70s %5d %8.2f  %s" % (r[0][:70], r[1], r[2], " ".join("%s=%s" % (w, v) for w, v in zip(want, r[3:]))))
