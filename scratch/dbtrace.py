"""Dump (kernel name, grid, duration us) in launch order from a rocprofv3 rocpd sqlite file."""
import sqlite3, sys
def rows(path):
    db = sqlite3.connect(path)
    t = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [x for x in t if 'kernel_dispatch' in x][0]
    ks = [x for x in t if 'kernel_symbol' in x][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    q = f"select s.kernel_name, d.grid_size_x, d.workgroup_size_x, d.grid_size_y, d.grid_size_z, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    return [(n.split('(')[0], gx // wx, gy, gz, (e - s) / 1e3) for n, gx, wx, gy, gz, s, e in db.execute(q)]
if __name__ == "__main__":
    for r in rows(sys.argv[1]):
        print("%-50s blocks=%6d y=%d z=%d  %8.1f us" % (r[0][:50], r[1], r[2], r[3], r[4]))
