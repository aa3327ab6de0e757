// Drives the reference's BROWSER game rules (the only place its full-row/column rule exists:
// src/gui/static/js/yin_yang_game.js:187-232 isValidMove, :338-384 checkRowColumnConstraint) under node, in the build
// container only.  The class text is read from the reference file at run time and evaluated in a vm sandbox (the YinYangGame
// class touches no DOM; the UI class and the DOMContentLoaded hooks below it are not evaluated).  Nothing of the reference
// is copied into this repository: the output is data (boards in, legal-move masks out).
//   node rowcol_from_js.js <reference js path> < boards.json > masks.json
const fs = require('fs');
const vm = require('vm');
const src = fs.readFileSync(process.argv[2], 'utf8');
const end = src.indexOf('class YinYangGameUI');
if (end < 0) throw new Error('YinYangGameUI marker not found');
const Game = vm.runInNewContext(src.slice(0, end) + '\n;YinYangGame', {});
const inp = JSON.parse(fs.readFileSync(0, 'utf8'));
const out = [];
for (const set of inp) {
    const { rows, cols, boards } = set;
    const masks = { p1: [], m1: [] };
    for (const flat of boards) {
        const g = new Game(rows, cols);
        for (let r = 0; r < rows; r++) for (let c = 0; c < cols; c++) g.board[r][c] = flat[r * cols + c];
        for (const [key, player] of [['p1', 1], ['m1', -1]]) {
            g.currentPlayer = player;
            const m = [];
            for (let r = 0; r < rows; r++) for (let c = 0; c < cols; c++) m.push(g.isValidMove(r, c) ? 1 : 0);
            masks[key].push(m);
            // isValidMove must leave the board as it found it
            for (let r = 0; r < rows; r++) for (let c = 0; c < cols; c++)
                if (g.board[r][c] !== flat[r * cols + c]) throw new Error('board mutated');
        }
    }
    out.push({ rows, cols, p1: masks.p1, m1: masks.m1 });
}
process.stdout.write(JSON.stringify(out));
