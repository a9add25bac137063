"""At which clock did the isolated launches of the counter passes run?  ("state in r04_pmc_* at which clock
the isolated PMC launches ran".)  Effective clock per dispatch = GRBM_GUI_ACTIVE / 8 XCDs / kernel duration, from the MFMA-busy
pass's counter CSV and its kernel trace (MI355X_MICROARCH.md, 'DVFS give-back').
usage: python tools/pmc_clock.py gpurun_out/final/pmc_mfma/mf_counter_collection.csv gpurun_out/final/pmc_mfma/mf_kernel_trace.csv > profiles/r04_pmc_clock.txt"""
import csv
import statistics
import sys

cc, kt = sys.argv[1:3]
dur = {int(r['Dispatch_Id']): int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in csv.DictReader(open(kt))}
rows = []
for r in csv.DictReader(open(cc)):
    if r['Counter_Name'] != 'GRBM_GUI_ACTIVE':
        continue
    d, k = int(r['Dispatch_Id']), r['Kernel_Name']
    if d in dur and dur[d] >= 300_000 and ('conv_' in k or 'upconv' in k):      # launches of >= 0.3 ms: the quotient reads high below
        rows.append((d, k.replace('(anonymous namespace)::', '').split('(')[0].replace('void ', '')[:56], dur[d] / 1e3, float(r['Counter_Value']) / 8.0 / dur[d]))
g = [r[3] for r in rows]
print('# effective clock of the ISOLATED launches of the rocprofv3 counter passes (tools/pmc_probe.py under --pmc; each launch follows')
print('# host-side set-up, i.e. starts from an idle board that ramps its clock): GRBM_GUI_ACTIVE / 8 / duration, launches >= 0.3 ms')
print(f'# {len(g)} launches: median {statistics.median(g):.3f} GHz, min {min(g):.3f}, max {max(g):.3f}.  In the training step the same kernels run')
print('# back to back at the sustained clock (rocm-smi 1.74-1.77 GHz during the bf16 legs, profiles/r03_clock_trace.txt; bare MFMA loops')
print('# 1.82 GHz in-kernel, profiles/r04_mfma_ceiling.txt): MFMA-busy FRACTIONS of these passes are per-cycle and comparable,')
print('# the DURATIONS of the isolated launches are not the in-step durations.')
print(f'# {"dispatch":>8s} {"kernel":56s} {"us":>9s} {"GHz":>6s}')
for d, k, us, f in rows:
    print(f'  {d:8d} {k:56s} {us:9.1f} {f:6.3f}')
