import sqlite3, sys
db=sys.argv[1]
c=sqlite3.connect(db)
rows=list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 from kernels group by name order by 3 desc"))
tot=sum(r[2] for r in rows)
print("%-75s %7s %12s %9s %9s %9s %6s"%("kernel","calls","total_us","avg_us","min_us","max_us","pct"))
for r in rows: print("%-75s %7d %12.1f %9.2f %9.2f %9.2f %6.2f"%(r[0][:75],r[1],r[2],r[3],r[4],r[5],100*r[2]/tot))
