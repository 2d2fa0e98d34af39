"""Per-level breakdown of a forest build from a rocprofv3 kernel trace (csv): which margin mode each level used."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
out, cur = [], None
for r in rows:
    n = r['Kernel_Name']; d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6
    if 'k_forest_create_split' in n:
        if cur is None or cur['seen']:
            cur = {'cs': 0, 'rows': [], 'node': 0.0, 'assign': 0.0, 'masks': 0.0, 'seen': False, 'tc': None, 'other': 0.0,
                   'dense': 0.0, 'exact': 0.0}
            out.append(cur)
        cur['cs'] += d
    elif cur is None:
        continue
    elif 'k_forest_margin_rows' in n or 'k_forest_screen_rows' in n:
        cur['seen'] = True; cur['rows'].append(d); cur['tc'] = max(int(re.search(r'<\d+, (\d+)[,>]', n).group(1)), cur['tc'] or 0)
    elif 'k_forest_dense_screen' in n or 'k_forest_dense_narrow' in n:
        cur['seen'] = True; cur['dense'] += d; cur['tc'] = 'mfma'
    elif 'k_forest_exact_pairs' in n:
        cur['seen'] = True; cur['exact'] += d
    elif 'k_forest_margin_f32' in n or 'k_forest_margin_bq' in n or 'k_forest_screen_node' in n:
        cur['seen'] = True; cur['node'] += d
    elif 'assign_node_of' in n: cur['assign'] += d
    elif 'masks_from_bytes' in n: cur['masks'] += d
    elif 'k_forest' in n: cur['other'] += d
tot = 0
for i, c in enumerate(out):
    t = sum(c['rows']) + c['node'] + c['assign'] + c['masks'] + c['cs'] + c['other'] + c['dense'] + c['exact']; tot += t
    if t > 1:
        print(f"L{i // 4:2d} tc={c['tc']} rows {len(c['rows'])}x{(sum(c['rows']) / max(1, len(c['rows']))):6.1f}={sum(c['rows']):7.1f}"
              f" dense {c['dense']:6.1f} exact {c['exact']:5.1f} node {c['node']:7.1f} assign {c['assign']:5.1f} masks {c['masks']:5.1f} split {c['cs']:5.1f} other {c['other']:5.1f} total {t:7.1f} ms")
print("sum", round(tot, 1), "ms")
