"""Compact one-character-per-instruction view of a kernel's loops in a hipcc -save-temps .s file
    python tools/asm_view.py file.s <mangled-name-substring> [min MFMAs per loop]
M mfma  E v_exp  P v_pk_*  A v_accvgpr  C v_cvt  x v_max  v other VALU  r ds_read  w ds_write  W s_waitcnt  n s_nop
B s_barrier  T global_atomic  G global/buffer load  S global store  s other scalar"""
import re
import sys


def code(lines):
    out = ''
    for l in lines:
        l = l.strip()
        if not l or l.startswith(';') or l.startswith('.'):
            continue
        op = l.split()[0]
        if op.startswith('v_mfma'): c = 'M'
        elif op.startswith('v_exp'): c = 'E'
        elif op.startswith('v_pk_'): c = 'P'
        elif op.startswith('v_accvgpr'): c = 'A'
        elif op.startswith('v_cvt'): c = 'C'
        elif op.startswith('v_max'): c = 'x'
        elif op.startswith('v_'): c = 'v'
        elif op.startswith('ds_read'): c = 'r'
        elif op.startswith('ds_write'): c = 'w'
        elif op.startswith('s_waitcnt'): c = 'W'
        elif op.startswith('s_nop'): c = 'n'
        elif op.startswith('s_barrier'): c = 'B'
        elif op.startswith('global_atomic'): c = 'T'
        elif op.startswith('global_load') or op.startswith('buffer_load'): c = 'G'
        elif op.startswith('global_store') or op.startswith('buffer_store'): c = 'S'
        elif op.startswith('s_'): c = 's'
        else: c = '?'
        out += c
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    minm = int(sys.argv[3]) if len(sys.argv) > 3 else 8
    src = open(path).read().split('\n')
    start = [i for i, l in enumerate(src) if re.match(r'^_Z\S*' + re.escape(name) + r'\S*:', l)]
    for st in start:
        end = next(i for i in range(st, len(src)) if 's_endpgm' in src[i])
        body = src[st:end + 1]
        print(src[st][:150])
        labels = {}
        for i, l in enumerate(body):
            m = re.match(r'^(\.LBB\d+_\d+):', l)
            if m:
                labels[m.group(1)] = i
        for i, l in enumerate(body):
            m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
            if m and m.group(1) in labels and labels[m.group(1)] < i:
                c = code(body[labels[m.group(1)]:i + 1])
                if c.count('M') >= minm:
                    print(f"  loop lines {labels[m.group(1)]}..{i}: {len(c)} instructions, {c.count('M')} MFMA, {c.count('P')} packed-f32, "
                          f"{c.count('E')} exp, {c.count('A')} accvgpr moves, {c.count('n')} s_nop, {c.count('W')} waitcnt")
                    for k in range(0, len(c), 110):
                        print('    ' + c[k:k + 110])


if __name__ == '__main__':
    main()
