import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'nte::' in r['Name'] and 'true, true' not in r['Name']:
        print("%-44s calls %4s avg %9.3f ms" % (r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e6))
