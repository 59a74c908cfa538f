import re, sys
for line in open(sys.argv[1]):
    m = re.match(r"\| `([^(<]+(?:<[^>]*>)?)[^`]*` \| (\d+) \| ([\d.]+) \| ([\d.]+) \|", line)
    if m and re.search(sys.argv[2], m.group(1)):
        print("%-60s calls %5s avg %8s us" % (m.group(1)[:60], m.group(2), m.group(4)))
