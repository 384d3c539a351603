from chacha_poseidon_consts import ChaCha8Rng, gen_range_u64
from grain_poseidon2_consts import poseidon2_rc12
P = 0xFFFFFFFF00000001
def poseidon_rc():
    rng = ChaCha8Rng(0)
    return [gen_range_u64(rng, P) for _ in range(360)]
RC = poseidon_rc()
CIRC = [17, 15, 41, 16, 2, 28, 13, 13, 39, 18, 34, 20]
DIAG = [8] + [0]*11
def poseidon(state):
    s = list(state)
    for r in range(30):
        s = [(s[i] + RC[12*r+i]) % P for i in range(12)]
        if r < 4 or r >= 26:
            s = [pow(x, 7, P) for x in s]
        else:
            s[0] = pow(s[0], 7, P)
        s = [(sum(s[(i + r2) % 12] * CIRC[i] for i in range(12)) + s[r2]*DIAG[r2]) % P for r2 in range(12)]
    return s
E1, IN, E2 = poseidon2_rc12()
D = [0xc3b6c08e23ba9300, 0xd84b5de94a324fb6, 0x0d0c371c5b35b84f, 0x7964f570e7188037, 0x5daf18bbd996604b,
0x6743bc47b9595257, 0x5528b9362c59bb70, 0xac45e25b7127b68b, 0xa2077d7dfbb606b5, 0xf3faac6faee378ae, 0x0c6388b51545e883, 0xd27dbb6944917b60]
M4 = [[5,7,1,3],[4,6,1,1],[1,3,5,7],[1,1,4,6]]
def ext(s):
    t = []
    for c in range(3):
        ch = s[4*c:4*c+4]
        t += [sum(M4[i][j]*ch[j] for j in range(4)) % P for i in range(4)]
    sums = [(t[i] + t[4+i] + t[8+i]) % P for i in range(4)]
    return [(t[i] + sums[i % 4]) % P for i in range(12)]
def poseidon2(state):
    s = ext(list(state))
    for r in range(4):
        s = [pow((s[i] + E1[r][i]) % P, 7, P) for i in range(12)]
        s = ext(s)
    for r in range(22):
        s[0] = pow((s[0] + IN[r]) % P, 7, P)
        tot = sum(s) % P
        s = [(s[i]*D[i] + tot) % P for i in range(12)]
    for r in range(4):
        s = [pow((s[i] + E2[r][i]) % P, 7, P) for i in range(12)]
        s = ext(s)
    return s
def hash_n_to_m_no_pad(perm, inp, m):
    st = [0]*12
    for i in range(0, len(inp), 8):
        ch = inp[i:i+8]
        st[:len(ch)] = ch
        st = perm(st)
    out = []
    while True:
        for x in st[:8]:
            out.append(x)
            if len(out) == m: return out
        st = perm(st)
if __name__ == "__main__":
    for name, perm in (("poseidon", poseidon), ("poseidon2", poseidon2)):
        for dst in (b"BLOCK_NUMBER",):
            print(name, dst, hash_n_to_m_no_pad(perm, list(dst), 4)[0])
    print("targets", 15542555334667826467, 17422912802427138938)
