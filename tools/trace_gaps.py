import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last 40% of rows (steady state)
n=len(rows); rows=rows[int(n*0.5):]
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows)
span=int(rows[-1]['End_Timestamp'])-int(rows[0]['Start_Timestamp'])
gaps=[int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']) for i in range(len(rows)-1)]
import statistics
print('kernels',len(rows),'busy us',busy/1e3,'span us',span/1e3,'gap total us',sum(gaps)/1e3,'median gap ns',statistics.median(gaps),'mean',statistics.mean(gaps))
short=[int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows]
print('median kernel ns',statistics.median(short),'min',min(short))
