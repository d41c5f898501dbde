'use strict';
/* node -r tests/js/mock_preload.js <script>: runs an unchanged JS host script (tests/js/c5_stream.js, ...) with the product addon replaced by
 * the oracle-backed mock (tests/js/mock_addon.js) — the script's HOST logic on a box without a GPU.  Test infrastructure. */
require(require('path').join(__dirname, 'mock_addon.js')).install();
