"""Does scikit-learn's k-max-heap (SKL/utils/_heap.pyx:6-85, heap_push) keep the same rows among EXACT ties when the rows
that end up strictly beyond the final k-th value are left out of the scan?  (VERDICT r3 item 8 proposed to replay tied queries
over "the candidates' tie class plus rows inside the certified bound" only.)  Random small sequences with ties, the heap
restated line by line: compare the surviving indices of the full sequence with those of the subsequence val <= final k-th value."""
import random


def heap_push(values, indices, size, val, idx):
    if val >= values[0]:
        return
    values[0], indices[0] = val, idx
    i = 0
    while True:
        ic1 = 2 * i + 1
        ic2 = ic1 + 1
        if ic1 >= size:
            break
        elif ic2 >= size:
            if values[ic1] > val:
                i_swap = ic1
            else:
                break
        elif values[ic1] >= values[ic2]:
            if val < values[ic1]:
                i_swap = ic1
            else:
                break
        else:
            if val < values[ic2]:
                i_swap = ic2
            else:
                break
        values[i], indices[i] = values[i_swap], indices[i_swap]
        i = i_swap
    values[i], indices[i] = val, idx


def run(seq, k):
    v, ix = [float("inf")] * k, [-1] * k
    for j, x in seq:
        heap_push(v, ix, k, x, j)
    return v, ix


random.seed(1)
bad = tot = 0
example = None
for _ in range(200_000):
    k = random.choice([2, 3, 4, 5, 6])
    seq = [(j, float(random.randint(0, 5))) for j in range(random.randint(k + 1, 14))]
    v, ix = run(seq, k)
    sub = [(j, x) for j, x in seq if x <= max(v)]
    _, ix2 = run(sub, k)
    tot += 1
    if sorted(ix) != sorted(ix2):
        bad += 1
        example = example or (k, seq, sorted(ix), sorted(ix2))
print(f"{bad} of {tot} random tie-heavy sequences keep DIFFERENT rows when the rows beyond the final k-th value are left out")
print("first example: k = %d, (index, value) sequence %s: full scan keeps %s, restricted scan keeps %s" % example)
