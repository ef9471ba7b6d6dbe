"""Re-attach the `traffic` / `*_from_profiles` keys of a stored bench line from the committed profiles (bench.attach_profiles):
tools/collect_profiles.sh writes `<tag>_bench_line.json` BEFORE the counter passes it would quote, so the line it stores carries
`traffic: null`.  Everything measured live in that line stays as it was.
    python tools/attach_profiles.py profiles/r6_bench_line.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
path = sys.argv[1]
line = json.loads(open(path).read().strip().splitlines()[-1])
md5 = json.load(open(os.path.join(ROOT, 'profiles', bench.PROFILE_TAG + '_pmc_traffic.json'))).get('lib_md5')      # the library of that collection run
bench.attach_profiles(line['roofline'], 256, True, md5)
if 'width128' in line and 'roofline' in line['width128']:
    bench.attach_profiles(line['width128']['roofline'], 128, False, md5)
open(path, 'w').write(json.dumps(line) + '\n')
r = line['roofline']
print('traffic', r['traffic'], 'frac', r['frac'], 'frac_from_profiles', r.get('frac_from_profiles'), 'match', r['profiles_match_this_build'])
