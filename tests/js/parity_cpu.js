'use strict';
/* CPU run of the drop-in sections of tests/js/parity_gpu.js — ccv.grayscale / detect_objects, getWhitebalance, camshift.Tracker incl. the
 * debug getters, the facetrackr WB -> VJ -> CS state machine, the headtrackr.Tracker loop (Smoother, headposition, status events) and the
 * main.js debug overlay — through the UNCHANGED facade headtrackr_amd/js/headtrackr.js + tracker.js, with the product addon replaced by
 * tests/js/mock_addon.js (the same entry points on the CPU oracle).  What this checks is the facade's HOST logic against the golden
 * vectors recorded from the reference JS; the kernels are checked on the GPU by the same file without the mock.
 *    node tests/js/parity_cpu.js <job.json>     (job.cpu_mock must be true)  */
const path = require('path');
require(path.join(__dirname, 'mock_addon.js')).install();
require(path.join(__dirname, 'parity_gpu.js'));
