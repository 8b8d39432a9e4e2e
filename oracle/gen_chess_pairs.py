"""Writes tests/golden/chess_conv_pairs.json: the (board side, move) <-> conv-policy index known answers that the
reference's own mapper tests hold (rust/kz-core/tests/mapper/chess/pairs.rs:16-356: queen distance / direction, knight
direction, promotions and under-promotions, en passant, castling, each for white and for black to move).  These are the
test VECTORS of that file (side to move, from, to, promotion, expected index), typed in as data; squares are written as
in the reference (A1..H8) and stored as rank * 8 + file."""
import json
import os


def sq(name):
    return (int(name[1]) - 1) * 8 + (ord(name[0].lower()) - ord("a"))


P = {None: 0, "Q": 1, "R": 2, "B": 3, "N": 4}
rows = []


def add(side, frm, to, promo, index, case):
    rows.append({"case": case, "white_to_move": side == "w", "from": sq(frm), "to": sq(to), "promotion": P[promo],
                 "conv_index": index})


d4 = sq("d4")
for d, to in enumerate(["a2", "a3", "a4", "a5", "a6", "a7", "a8"]):
    add("w", "a1", to, None, d * 64, "queen_distance_white")
for d, to in enumerate(["a7", "a6", "a5", "a4", "a3", "a2", "a1"]):
    add("b", "a8", to, None, d * 64, "queen_distance_black")
for k, to in enumerate(["d5", "e5", "e4", "e3", "d3", "c3", "c4", "c5"]):
    add("w", "d4", to, None, k * 7 * 64 + d4, "queen_direction_white")
for k, to in enumerate(["d4", "e4", "e5", "e6", "d6", "c6", "c5", "c4"]):
    add("b", "d5", to, None, k * 7 * 64 + d4, "queen_direction_black")
for k, to in enumerate(["e6", "f5", "f3", "e2", "c2", "b3", "b5", "c6"]):
    add("w", "d4", to, None, (56 + k) * 64 + d4, "knight_direction_white")
for k, to in enumerate(["e3", "f4", "f6", "e7", "c7", "b6", "b4", "c3"]):
    add("b", "d5", to, None, (56 + k) * 64 + d4, "knight_direction_black")

c = "white_potential_promotions"
add("w", "f6", "f8", None, (0 * 7 + 1) * 64 + sq("f6"), c)
add("w", "g7", "g8", None, (0 * 7 + 0) * 64 + sq("g7"), c)
add("w", "g6", "f8", None, 63 * 64 + sq("g6"), c)
add("w", "g6", "h8", None, 56 * 64 + sq("g6"), c)
b7 = sq("b7")
add("w", "b7", "a8", "Q", (7 * 7 + 0) * 64 + b7, c)
add("w", "b7", "b8", "Q", (0 * 7 + 0) * 64 + b7, c)
add("w", "b7", "c8", "Q", (1 * 7 + 0) * 64 + b7, c)
for piece, base in (("R", 64), ("B", 65), ("N", 66)):
    for j, to in enumerate(["a8", "b8", "c8"]):
        add("w", "b7", to, piece, (base + 3 * j) * 64 + b7, c)

c = "black_potential_promotions"  # indices from the point of view of black
add("b", "c3", "c1", None, (0 * 7 + 1) * 64 + sq("c6"), c)
add("b", "b2", "b1", None, (0 * 7 + 0) * 64 + sq("b7"), c)
add("b", "b3", "c1", None, 56 * 64 + sq("b6"), c)
add("b", "b3", "a1", None, 63 * 64 + sq("b6"), c)
g7 = sq("g7")
add("b", "g2", "f1", "Q", (7 * 7 + 0) * 64 + g7, c)
add("b", "g2", "g1", "Q", (0 * 7 + 0) * 64 + g7, c)
add("b", "g2", "h1", "Q", (1 * 7 + 0) * 64 + g7, c)
for piece, base in (("R", 64), ("B", 65), ("N", 66)):
    for j, to in enumerate(["f1", "g1", "h1"]):
        add("b", "g2", to, piece, (base + 3 * j) * 64 + g7, c)

add("w", "c5", "b6", None, (7 * 7 + 0) * 64 + sq("c5"), "en_passant")
add("b", "b4", "c3", None, (1 * 7 + 0) * 64 + sq("b5"), "en_passant")
e1 = sq("e1")
add("w", "e1", "g1", None, (2 * 7 + 1) * 64 + e1, "castles")
add("w", "e1", "c1", None, (6 * 7 + 1) * 64 + e1, "castles")
add("b", "e8", "g8", None, (2 * 7 + 1) * 64 + e1, "castles")
add("b", "e8", "c8", None, (6 * 7 + 1) * 64 + e1, "castles")

out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "chess_conv_pairs.json")
json.dump(rows, open(out, "w"), indent=0)
# the same vectors as plain lines for the C++ host test: white_to_move from to promotion conv_index
with open(out.replace(".json", ".txt"), "w") as f:
    for r in rows:
        f.write(f"{int(r['white_to_move'])} {r['from']} {r['to']} {r['promotion']} {r['conv_index']}\n")
print(len(rows), "pairs ->", out)
