"""extract one kernel's body from a `hipcc -S --cuda-device-only` listing.  usage: kernel_asm.py FILE.s SUBSTRING > out.s"""
import re, sys
s = open(sys.argv[1]).read()
m = re.search(r'^(\S*' + re.escape(sys.argv[2]) + r'\S*):', s, re.M)
end = re.compile(r'^\.Lfunc_end\d+:', re.M).search(s, m.end()).start()
sys.stdout.write(s[m.start():end])
